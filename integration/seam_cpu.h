/* seam_cpu.h -- REFERENCE-SIDE MEASUREMENT AID (integration/): host CPU time per pipeline stage.
 *
 * With SVT_HIP_SEAM_CPU_STATS=<file> every stage entry the seams wrap -- the reference's own function when the seam is off, the device stage call when it is on --
 * is bracketed with the calling thread's CPU clock (CLOCK_THREAD_CPUTIME_ID: user + system time of that thread only, spinning included) and the sums are written at
 * exit: `<stage>_cpu_ms N` and `<stage>_calls N` for me, tf, tpl, dlf, cdef, lr.  This is the accounting VERDICT r3 asks for: what each stage of SURVEY 8 costs the
 * HOST with the reference's own kernels (the AVX2 / AVX-512 builds, no seam) and what it costs with the stage on the MI355X.  Unset: one predictable branch per call. */
#ifndef SVT_HIP_SEAM_CPU_H
#define SVT_HIP_SEAM_CPU_H
#include <time.h>
enum { SEAM_CPU_ME, SEAM_CPU_TF, SEAM_CPU_TPL, SEAM_CPU_DLF, SEAM_CPU_CDEF, SEAM_CPU_LR, SEAM_CPU_STAGES };
int  svt_hip_seam_cpu_on(void);                              /* integration/enc_handle_binding.c */
void svt_hip_seam_cpu_add(int stage, unsigned long long ns);
static inline unsigned long long seam_cpu_ns(void) {
    struct timespec t_;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t_);
    return (unsigned long long)t_.tv_sec * 1000000000ull + (unsigned long long)t_.tv_nsec;
}
/* Timing experiment ("10-bit preset 8"): SVT_HIP_SEAM_DELAY_US=<n> makes the picture-level CDEF and loop-restoration stage entries (seam_test_delay() in their seams) sleep n microseconds first -- with NO seam on, i.e. in the
 * plain reference encoder.  It answers whether a bitstream difference seen with a device stage comes from the stage's RESULTS or from the stage merely taking a
 * different time than the reference's own function (a latent race of the reference that a timing change exposes). */
#include <stdlib.h>
#include <unistd.h>
/* A setting read from the environment on first use.  Every thread that gets there computes the same value, but several may get there together: relaxed atomics keep
 * that a defined program (found by ThreadSanitizer on the seams' lazy statics: profiles/r05_tsan_seams.txt). */
#define SEAM_ENV_ONCE(var, expr)                                                \
    static int var##_cache_ = -1;                                               \
    int        var          = __atomic_load_n(&var##_cache_, __ATOMIC_RELAXED); \
    if (var < 0) { var = (expr); __atomic_store_n(&var##_cache_, var, __ATOMIC_RELAXED); }
static inline void seam_test_delay(void) {
    SEAM_ENV_ONCE(us_, (getenv("SVT_HIP_SEAM_DELAY_US") ? abs(atoi(getenv("SVT_HIP_SEAM_DELAY_US"))) : 0));
    if (us_ > 0) usleep((useconds_t)us_);
}
#define SEAM_CPU_BEGIN() const int seam_cpu_on_ = svt_hip_seam_cpu_on(); const unsigned long long seam_cpu_t0_ = seam_cpu_on_ ? seam_cpu_ns() : 0
#define SEAM_CPU_END(stage) do { if (seam_cpu_on_) svt_hip_seam_cpu_add(stage, seam_cpu_ns() - seam_cpu_t0_); } while (0)
#endif
