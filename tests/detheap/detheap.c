/* detheap.c -- TEST HARNESS: an LD_PRELOAD shim that zero-fills every heap block at allocation, so that a read of memory nobody wrote returns the same thing in every run.
 *
 * The reference encoder reads a handful of heap arrays before writing them (MemorySanitizer on the plain C encoder, profiles/r05_reference_msan_10bit.txt: e.g.
 * b64_me_qindex of pcs.c:539 in svt_aom_get_me_qindex, md_rate_estimation.c:1035).  Fresh pages from the kernel are zero, recycled heap chunks are not, so what such a
 * read returns depends on the allocation history.  tools/enc_identity.py preloads this into BOTH encodes of a 10-bit comparison (together with SVT_HIP_TEST_SCRUB_PCS=1,
 * integration/pic_manager_process_seam.c, which covers the pool objects that are re-used after their first picture) and then demands first-attempt equality.
 * Never loaded outside the tests.   gcc -O2 -fPIC -shared -o libdetheap.so detheap.c   (tests/test_encoder_identity.py builds it next to the emulator) */
#define _GNU_SOURCE
#include <errno.h>
#include <stddef.h>
#include <string.h>

extern void *__libc_malloc(size_t);
extern void *__libc_memalign(size_t, size_t);
extern void *__libc_realloc(void *, size_t);
extern size_t malloc_usable_size(void *);

void *malloc(size_t n) {
    void *p = __libc_malloc(n);
    if (p)
        memset(p, 0, n);
    return p;
}
void *memalign(size_t a, size_t n) {
    void *p = __libc_memalign(a, n);
    if (p)
        memset(p, 0, n);
    return p;
}
void *aligned_alloc(size_t a, size_t n) {
    return memalign(a, n);
}
int posix_memalign(void **out, size_t a, size_t n) {
    if (a < sizeof(void *) || (a & (a - 1)))
        return EINVAL;
    void *p = memalign(a, n);
    if (!p)
        return ENOMEM;
    *out = p;
    return 0;
}
void *realloc(void *old, size_t n) { /* the grown tail is zero-filled as well */
    const size_t had = old ? malloc_usable_size(old) : 0;
    void        *p   = __libc_realloc(old, n);
    if (p && n > had)
        memset((char *)p + had, 0, n - had);
    return p;
}
