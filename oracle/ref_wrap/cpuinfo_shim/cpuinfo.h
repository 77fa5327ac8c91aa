/* cpuinfo.h -- TEST / BASELINE INFRASTRUCTURE: stand-in for the pytorch/cpuinfo library the reference's build FETCHES (it is not vendored under
 * /root/reference and there is no network here).  Only used by oracle/Makefile's `enc_avx2` target, which compiles the reference's common_dsp_rtcd.c with
 * -DHAVE_CPUINFO=1 so that svt_aom_get_cpu_flags() (common_dsp_rtcd.c:97-140) reports what the host CPU really supports; the answers come from the
 * compiler's __builtin_cpu_supports (CPUID + XGETBV, i.e. it also checks that the OS saves the AVX state). */
#ifndef SVT_REF_CPUINFO_SHIM_H
#define SVT_REF_CPUINFO_SHIM_H
#include <stdbool.h>
static inline bool cpuinfo_initialize(void) { __builtin_cpu_init(); return true; }
#define SHIM(name, feat) static inline bool cpuinfo_has_x86_##name(void) { return __builtin_cpu_supports(feat) != 0; }
SHIM(mmx, "mmx") SHIM(sse, "sse") SHIM(sse2, "sse2") SHIM(sse3, "sse3") SHIM(ssse3, "ssse3") SHIM(sse4_1, "sse4.1") SHIM(sse4_2, "sse4.2")
SHIM(avx, "avx") SHIM(avx2, "avx2")
SHIM(avx512f, "avx512f") SHIM(avx512dq, "avx512dq") SHIM(avx512cd, "avx512cd") SHIM(avx512bw, "avx512bw") SHIM(avx512vl, "avx512vl")
SHIM(avx512ifma, "avx512ifma") SHIM(avx512vbmi, "avx512vbmi") SHIM(avx512vpopcntdq, "avx512vpopcntdq") SHIM(avx512vnni, "avx512vnni")
SHIM(avx512vbmi2, "avx512vbmi2") SHIM(avx512bitalg, "avx512bitalg") SHIM(gfni, "gfni") SHIM(vpclmulqdq, "vpclmulqdq") SHIM(vaes, "vaes")
#undef SHIM
#endif
