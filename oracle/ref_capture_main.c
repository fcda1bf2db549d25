/*
 * ref_capture_main.c - TEST INFRASTRUCTURE.  The executable half of rt_ref_capture: it defines render(), which
 * interposes the definition inside librt_ref_capture.so (the unmodified reference, compiled position-independent
 * together with oracle/ref_harness.c), so that the world a reference scene function builds reaches the harness's
 * per-pixel loop instead of the reference's fixed-size one.
 */
int rt_harness_main(int argc, char** argv);
void rt_capture_render(void* c, void* w);

void render(void* c, void* w) { rt_capture_render(c, w); }

int main(int argc, char** argv) { return rt_harness_main(argc, argv); }
