/*
 * ref_dropin_lib.c - TEST INFRASTRUCTURE: the UNMODIFIED reference plus the
 * maintainer's binding (raytracer_amd/host/rt_reference_binding.c) as one
 * position-independent shared object.  Built -fPIC, so the reference's scene
 * functions reach render() through the PLT and the executable that loads this
 * library (ref_dropin_main.c) can supply its own render(): plain ELF symbol
 * interposition, no edit of the reference source.
 */
#define main rt_reference_main
#include "RayTracer.c"
#undef main
#include "rt_reference_binding.c"
