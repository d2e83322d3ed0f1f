/*
 * ref_main_shim.cpp — the reference's main.cpp as a library (TEST INFRASTRUCTURE ONLY; built only in the container that has /root/reference).
 *
 * main.cpp:25-60 holds the reference's reader and writer of the "REF_00.01" image files its RMSE check compares a render with.  They are ordinary
 * functions of that translation unit, so the file itself is #included from where it lies (-I/root/reference) with `main` renamed - nothing of it is
 * copied - and two C wrappers call loadReference / saveReference (the latter exists under STORE_REFERENCE, main.cpp:14,24-34).  main() itself is never
 * called (it needs the staircase asset files, which the snapshot does not hold); its calls of initRenderer / runRenderer / cleanupRenderer
 * (kernels.h:6-8) resolve, as for any host of the drop-in boundary, to librt_mi355x.so, which oracle/Makefile links this library against.
 * Pins cuda-raytracing-optimized_amd/host/rt_harness.cpp (rtSaveReference / rtLoadReference): tests/test_oracle_vs_ref.py.
 */
#define STORE_REFERENCE
#define main ref_reference_main
#include "main.cpp"
#undef main

extern "C" {

/* loadReference(file, reference, nx, ny): 0 when it returned true.  Its complaints go to std::cerr: captured, keep the test log quiet. */
int ref_load_reference(const char* path, float* reference, int nx, int ny) {
    std::stringstream sink;
    std::streambuf* old = std::cerr.rdbuf(sink.rdbuf());
    const bool ok = std::ifstream(path).good() && loadReference(path, reinterpret_cast<vec3*>(reference), nx, ny);
    std::cerr.rdbuf(old);
    return ok ? 0 : -1;
}

void ref_save_reference(const char* path, int nx, int ny, const float* colors) {
    saveReference(path, nx, ny, reinterpret_cast<const vec3*>(colors));
}

}  // extern "C"
