/* enc_handle_binding.c -- REFERENCE-SIDE BINDING (what a maintainer of the reference adds; built into the reference encoder by oracle/Makefile for the identity / fps runs): the reference encoder with the binding of INTEGRATION.md §1.
 *
 * This translation unit IS Source/Lib/Globals/enc_handle.c of the reference (included below where it lies; nothing is copied)
 * plus the few lines a maintainer adds after enc_handle.c:1444-1445, where svt_av1_enc_init() assigns the run-time dispatch
 * table single-threaded, before init_fn_ptr() copies SAD pointers into svt_aom_mefn_ptr[] and before any worker thread exists:
 *
 *     svt_aom_setup_common_rtcd_internal(flags);
 *     svt_aom_setup_rtcd_internal(flags);
 *   + if (getenv("SVT_HIP")) { svt_hip_init(atoi(getenv("SVT_HIP"))); svt_hip_setup_rtcd(0); }
 *
 * The insertion is made by giving the second call a macro name for the duration of the #include.  The HIP library is
 * dlopen()ed (RTLD_GLOBAL, so its weak references to the RTCD pointer globals bind to libSvtAv1Enc's), which keeps this
 * encoder build free of any link-time dependency on ROCm: with SVT_HIP unset it is the plain C-only reference encoder
 * (the `--asm c` leg of the bitstream-identity check, .gitlab/workflows/linux/.gitlab-ci.yml:351-367).
 *
 * Environment: SVT_HIP=<device index> enables the hook; SVT_HIP_LIB=<path> overrides the library (the CPU test-suite points
 * it at the lock-step emulator build); SVT_HIP_ONLY=<comma list of pointer-name prefixes> and SVT_HIP_COUNT=<file> are read
 * by svt_hip_setup_rtcd itself (csrc/rtcd_hook.hip).
 */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>

#include "aom_dsp_rtcd.h" /* declares svt_aom_setup_rtcd_internal before the macro below exists */

static void svt_aom_setup_rtcd_then_hip(EbCpuFlags flags) {
    svt_aom_setup_rtcd_internal(flags);
    const char *dev = getenv("SVT_HIP");
    if (!dev)
        return;
    const char *path = getenv("SVT_HIP_LIB");
    void       *h    = dlopen(path ? path : "libsvtav1_hip.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        fprintf(stderr, "SVT_HIP: cannot load the HIP variant: %s\n", dlerror());
        abort(); /* no silent CPU fallback: the identity check must not pass on the C kernels */
    }
    int (*init)(int)               = (int (*)(int))dlsym(h, "svt_hip_init");
    int (*setup)(unsigned long long) = (int (*)(unsigned long long))dlsym(h, "svt_hip_setup_rtcd");
    if (!init || !setup || init(atoi(dev)) != 0) {
        fprintf(stderr, "SVT_HIP: svt_hip_init(%s) failed\n", dev);
        abort();
    }
    int n = setup((unsigned long long)flags);
    fprintf(stderr, "SVT_HIP: %d dispatch pointers now select the HIP variant\n", n);
}

#define svt_aom_setup_rtcd_internal(flags) svt_aom_setup_rtcd_then_hip(flags)
#include "enc_handle.c" /* resolves through -I$(REF)/Source/Lib/Globals */
