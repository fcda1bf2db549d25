/*
 * rt_reference_dropin_lib.c - the UNMODIFIED reference plus the maintainer's binding (rt_reference_binding.c) as
 * one position-independent shared object: `make dropin REFERENCE=<dir of RayTracer.c>`.  The reference source is
 * compiled where it lies; nothing of it is copied.  Built -fPIC, so the reference's scene functions reach render()
 * through the PLT and the executable that loads this library (rt_reference_dropin_main.c) can supply its own
 * render(): plain ELF symbol interposition, no edit of the reference source.
 */
#define main rt_reference_main
#include "RayTracer.c"
#undef main
#include "rt_reference_binding.c"
