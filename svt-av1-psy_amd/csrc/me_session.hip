// me_session.hip -- the open-loop ME stage as a per-picture service over HOST pictures (the reference's contract, SURVEY 8b: the caller owns
// host buffers; me_process.c hands one picture at a time to motion_estimation.c).  A session keeps the last `ring` luma planes resident in HBM,
// so a picture is uploaded once (as the source) and then serves as a reference for later pictures without crossing PCIe again.  Every
// submission runs on its own HIP stream: upload (copy engine) -> descriptor build -> me_fullpel search -> result download, so the copies of
// picture N + 1 overlap the search of picture N.  Dependencies are events: a search waits for the uploads of its references, an upload into a
// ring slot waits for every search that may still read the plane it replaces.
#include "svt_hip_common.h"
#include "../../include/svtav1_hip.h"

#include <vector>

namespace {

struct Slot {
    hipStream_t         st    = nullptr;
    hipEvent_t          done  = nullptr;
    SvtHipMeSearchDesc* descs = nullptr;
    uint32_t *          sad = nullptr, *mv = nullptr;
    void*               ws    = nullptr;
    uint8_t*            fmt   = nullptr; // formatted results: do_ref | total | cand | pad | mv | stats (see fmt_layout)
    uint8_t*            hme   = nullptr; // stage form: per level sad (u64) and centres (2 x i16) of every item, then final centre / sad, then workspace
    bool                busy  = false;
    std::vector<int>    reads;           // ring entries the slot's picture in flight reads (its source and references)
};
struct Session {
    int      device = 0;                 // the GPU the ring, the slots and every launch of this session live on (svt_hip_me_session_create_on)
    uint32_t width, height, stride, org_x, org_y, rows, ring, max_refs, sbs;
    size_t   plane_bytes, ws_bytes;
    uint8_t* sb_size = nullptr;          // [sbs][2] B64Geom width, height
    uint32_t max_cand = 0;               // bound over every (l0, l1) split of max_refs
    uint8_t* planes = nullptr;           // ring x plane_bytes
    std::vector<int64_t>    ids;         // picture id resident in ring slot r (-1 = empty)
    std::vector<hipEvent_t> uploaded;    // upload of ring slot r finished
    std::vector<Slot>       slots;
    uint32_t next_ring = 0, next_slot = 0;
    // stage form (svt_hip_me_session_enable_stage): quarter and sixteenth planes of every ring entry, made on the device right after the upload
    bool     stage = false;
    uint32_t lvl_pad[2] = {0, 0}, lvl_stride[2] = {0, 0}, lvl_rows[2] = {0, 0}, max_regions = 0, max_area_w = 0, max_area_h = 0;
    size_t   lvl_bytes[2] = {0, 0}, hme_items = 0, int_ws = 0;
    uint8_t* lvl_planes[2] = {nullptr, nullptr}; // [0] quarter, [1] sixteenth: ring x lvl_bytes
};

// all 64x64 SBs of the picture against n_refs resident planes, search area centred on the co-located block (search centre (0, 0)), exactly the
// operands open_loop_me_fullpel_search_sblock receives (motion_estimation.c:781); item order = reference-major like me_descs_for_frame
__global__ void me_build_descs_kernel(SvtHipMeSearchDesc* descs, uint32_t sbs_x, uint32_t sbs, uint32_t n_refs, uint32_t stride, uint32_t org_x, uint32_t org_y,
                                      unsigned long long src_off, const unsigned long long* ref_offs, int area_w, int area_h) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sbs * n_refs) return;
    const uint32_t r = i / sbs, sb = i - r * sbs, sx = sb % sbs_x, sy = sb / sbs_x;
    const int      xo = -(area_w >> 1), yo = -(area_h >> 1);
    const unsigned long long px = org_x + sx * 64, py = org_y + sy * 64;
    SvtHipMeSearchDesc d;
    d.src_off = src_off + py * stride + px;
    d.ref_off = ref_offs[r] + (unsigned long long)((long long)py + yo) * stride + (unsigned long long)((long long)px + xo);
    d.src_stride = stride; d.ref_stride = stride;
    d.x_search_area_origin = (int16_t)xo; d.y_search_area_origin = (int16_t)yo;
    d.search_area_width = (uint16_t)area_w; d.search_area_height = (uint16_t)area_h;
    descs[i] = d;
}

// B64Geom width / height of every SB: MIN(64, aligned_width - org_x) with the picture size rounded up to a multiple of 8 (pcs.c:1496-1518;
// motion_estimation.c:3093-3100)
__global__ void me_sb_size_kernel(uint8_t* sb_size, uint32_t sbs_x, uint32_t sbs, uint32_t width, uint32_t height) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sbs) return;
    const uint32_t aw = (width + 7) & ~7u, ah = (height + 7) & ~7u, x = (i % sbs_x) * 64, y = (i / sbs_x) * 64;
    sb_size[2 * i]     = (uint8_t)(aw - x < 64 ? aw - x : 64);
    sb_size[2 * i + 1] = (uint8_t)(ah - y < 64 ? ah - y : 64);
}
struct FmtLayout { size_t do_ref, total, cand, mv, stats, bytes; };
inline FmtLayout fmt_layout(uint32_t sbs, uint32_t n_pus, uint32_t max_refs, uint32_t max_cand) {
    FmtLayout L;
    L.do_ref = 0;
    L.total  = svthip::align_up((size_t)sbs * 8, 16);
    L.cand   = svthip::align_up(L.total + (size_t)sbs * n_pus, 16);
    L.mv     = svthip::align_up(L.cand + (size_t)sbs * n_pus * max_cand, 16);
    L.stats  = svthip::align_up(L.mv + (size_t)sbs * n_pus * max_refs * 4, 16);
    L.bytes  = L.stats + (size_t)sbs * sizeof(SvtHipMeSbStats);
    return L;
}

} // namespace

extern "C" {

void* svt_hip_host_alloc(size_t bytes) {
    SVT_HIP_ENTRY_TRY
    svthip::ensure_device();
    void* p = nullptr;
    HIP_CHECK(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault));
    return p;
    SVT_HIP_ENTRY_CATCH(nullptr)
}
void svt_hip_host_free(void* p) {
    SVT_HIP_ENTRY_TRY if (p) HIP_CHECK(hipHostFree(p));     SVT_HIP_ENTRY_CATCH((void)0)
}

void* svt_hip_me_session_create(uint32_t width, uint32_t height, uint32_t stride, uint32_t org_x, uint32_t org_y, uint32_t rows, uint32_t ring_planes,
                                uint32_t max_refs, uint32_t max_area_width, uint32_t max_area_height, uint32_t n_slots) {
    SVT_HIP_ENTRY_TRY
    svthip::ensure_device();
    return svt_hip_me_session_create_on(svthip::current_device(), width, height, stride, org_x, org_y, rows, ring_planes, max_refs, max_area_width, max_area_height,
                                        n_slots);
    SVT_HIP_ENTRY_CATCH(nullptr)
}
void* svt_hip_me_session_create_on(int device, uint32_t width, uint32_t height, uint32_t stride, uint32_t org_x, uint32_t org_y, uint32_t rows, uint32_t ring_planes,
                                   uint32_t max_refs, uint32_t max_area_width, uint32_t max_area_height, uint32_t n_slots) {
    SVT_HIP_ENTRY_TRY
    if (device < 0 || device >= svt_hip_device_count() || device >= svthip::MAX_DEVICES) return nullptr; // (the per-device arenas are MAX_DEVICES wide)
    svthip::DeviceGuard guard(device);
    Session* s = new Session;
    s->device = device;
    s->width = width; s->height = height; s->stride = stride; s->org_x = org_x; s->org_y = org_y; s->rows = rows;
    s->ring = ring_planes < 2 ? 2 : ring_planes; s->max_refs = max_refs ? max_refs : 1;
    s->sbs = ((width + 63) / 64) * ((height + 63) / 64);
    s->plane_bytes = (size_t)stride * rows;
    HIP_CHECK(hipMalloc((void**)&s->planes, s->plane_bytes * s->ring));
    s->ids.assign(s->ring, -1);
    s->uploaded.resize(s->ring);
    for (auto& e : s->uploaded) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const size_t n = (size_t)s->sbs * s->max_refs;
    s->ws_bytes    = svt_hip_me_fullpel_search_workspace((uint32_t)n, max_area_width, max_area_height);
    s->max_cand = 2 * s->max_refs + (s->max_refs * s->max_refs + 3) / 4 + 1; // >= refs + l0 * l1 + (l0 - 1) + 1 for every split (pcs.c:91-96)
    HIP_CHECK(hipMalloc((void**)&s->sb_size, (size_t)s->sbs * 2));
    hipLaunchKernelGGL(me_sb_size_kernel, dim3((s->sbs + 255) / 256), dim3(256), 0, 0, s->sb_size, (width + 63) / 64, s->sbs, width, height);
    SVT_LAUNCH_CHECK();
    HIP_CHECK(hipStreamSynchronize(0));
    s->slots.resize(n_slots ? n_slots : 2);
    for (auto& sl : s->slots) {
        HIP_CHECK(hipStreamCreateWithFlags(&sl.st, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        HIP_CHECK(hipMalloc((void**)&sl.descs, n * sizeof(SvtHipMeSearchDesc) + s->max_refs * 8));
        HIP_CHECK(hipMalloc((void**)&sl.sad, n * SVT_HIP_ME_NUM_BLOCKS * 4));
        HIP_CHECK(hipMalloc((void**)&sl.mv, n * SVT_HIP_ME_NUM_BLOCKS * 4));
        if (s->ws_bytes) HIP_CHECK(hipMalloc(&sl.ws, s->ws_bytes));
        HIP_CHECK(hipMalloc((void**)&sl.fmt, fmt_layout(s->sbs, SVT_HIP_ME_NUM_BLOCKS, s->max_refs, s->max_cand).bytes));
    }
    return s;
    SVT_HIP_ENTRY_CATCH(nullptr)
}

void svt_hip_me_session_destroy(void* session) {
    SVT_HIP_ENTRY_TRY
    Session* s = (Session*)session;
    if (!s) return;
    svthip::DeviceGuard guard(s->device);
    if (!s) return;
    svthip::ensure_device();
    for (auto& sl : s->slots) {
        HIP_CHECK(hipStreamSynchronize(sl.st));
        HIP_CHECK(hipStreamDestroy(sl.st));
        HIP_CHECK(hipEventDestroy(sl.done));
        HIP_CHECK(hipFree(sl.descs)); HIP_CHECK(hipFree(sl.sad)); HIP_CHECK(hipFree(sl.mv));
        if (sl.ws) HIP_CHECK(hipFree(sl.ws));
        HIP_CHECK(hipFree(sl.fmt));
        if (sl.hme) HIP_CHECK(hipFree(sl.hme));
    }
    for (int k = 0; k < 2; k++)
        if (s->lvl_planes[k]) HIP_CHECK(hipFree(s->lvl_planes[k]));
    HIP_CHECK(hipFree(s->sb_size));
    for (auto& e : s->uploaded) HIP_CHECK(hipEventDestroy(e));
    HIP_CHECK(hipFree(s->planes));
    delete s;
    SVT_HIP_ENTRY_CATCH((void)0)
}

static int me_session_submit(void* session, int64_t pic_id, const uint8_t* plane_host, const int64_t* ref_ids, uint32_t n_refs, uint32_t area_w,
                             uint32_t area_h, int sub_sad, uint32_t* best_sad_host, uint32_t* best_mv_host, const SvtHipMeResultsParams* fmt,
                             const SvtHipMeResultsHost* out, const SvtHipMeStageParams* stage = nullptr) {
    Session* s = (Session*)session;
    svthip::DeviceGuard guard(s->device); // an encoder worker thread that did not create the session binds to the session's device here
    if (n_refs > s->max_refs) return -2;
    if (stage && (!s->stage || n_refs > 8 || (uint32_t)stage->num_hme_sa_w * stage->num_hme_sa_h > s->max_regions ||
                  stage->num_hme_sa_w == 0 || stage->num_hme_sa_h == 0))
        return -5;
    if (fmt && ((uint32_t)fmt->num_of_ref_pic_to_search[0] + fmt->num_of_ref_pic_to_search[1] != n_refs || n_refs == 0 || fmt->max_refs > s->max_refs ||
                fmt->max_cand > s->max_cand))
        return -4;
    if (stage && n_refs) { // every refusal happens here, before anything is enqueued or any session state changes
        if (stage->sr_adjustment > 2) return -5;
        if (stage->hme_levels > 3) return -5;
        if (stage->prehme_enabled && (stage->num_hme_sa_w != 2 || stage->num_hme_sa_h != 2)) return -5; // get_worst_quadrant is written for 2 x 2 regions
        if ((uint32_t)stage->results.num_of_ref_pic_to_search[0] + stage->results.num_of_ref_pic_to_search[1] != n_refs) return -4;
        SvtHipMeIntegerSearchParams V; // the fields svt_hip_me_integer_search_workspace / me_int_max_area read
        memset(&V, 0, sizeof(V));
        V.sbs_x = (s->width + 63) / 64; V.sbs_y = s->sbs / V.sbs_x; V.n_refs = n_refs;
        V.sa_min_width = stage->me_sa_min_width; V.sa_min_height = stage->me_sa_min_height; V.sa_max_width = stage->me_sa_max_width; V.sa_max_height = stage->me_sa_max_height;
        V.mv_adj_enabled = stage->mv_adj_enabled; V.mv_adj_sa_multiplier = stage->mv_adj_sa_multiplier; V.me_8x8_var_enabled = stage->me_8x8_var_enabled;
        for (uint32_t k = 0; k < n_refs; k++) V.dist[k] = stage->dist[k];
        if (svt_hip_me_integer_search_workspace(&V) > s->int_ws) return -5;        // areas larger than the session was sized for (variance-probe enlargement included)
    }
    const int si = (int)s->next_slot;
    Slot&     sl = s->slots[si];
    if (sl.busy) { HIP_CHECK(hipEventSynchronize(sl.done)); sl.busy = false; } // the slot's previous picture (results already fetched or abandoned)
    // the source plane: resident already, or uploaded into the next ring slot
    int src_r = -1;
    for (uint32_t r = 0; r < s->ring; r++)
        if (s->ids[r] == pic_id) src_r = (int)r;
    std::vector<int> ref_r(n_refs);
    for (uint32_t k = 0; k < n_refs; k++) {
        ref_r[k] = -1;
        for (uint32_t r = 0; r < s->ring; r++)
            if (s->ids[r] == ref_ids[k]) ref_r[k] = (int)r;
        if (ref_r[k] < 0) return -1; // reference not resident (evicted or never submitted)
    }
    if (src_r < 0) {
        if (!plane_host) return -1;
        // a freed entry first (svt_hip_me_session_invalidate left id -1 there: re-uploading an invalidated picture must not push a live reference out of the
        // ring), otherwise the next ring slot that is not one of this picture's references
        for (uint32_t r = 0; r < s->ring && src_r < 0; r++)
            if (s->ids[r] == -1) src_r = (int)r;
        for (uint32_t tries = 0; tries < s->ring && src_r < 0; tries++) {
            const uint32_t r = (s->next_ring + tries) % s->ring;
            bool used = false;
            for (uint32_t k = 0; k < n_refs; k++) used |= ref_r[k] == (int)r;
            if (!used) { src_r = (int)r; s->next_ring = (r + 1) % s->ring; break; }
        }
        if (src_r < 0) return -3; // ring smaller than n_refs + 1
        for (auto& other : s->slots) { // a picture in flight that still reads the ring entry being replaced has to finish first; the others overlap
            bool reads = false;
            for (int r : other.reads) reads |= r == src_r;
            if (other.busy && reads) HIP_CHECK(hipStreamWaitEvent(sl.st, other.done, 0));
        }
        HIP_CHECK(hipMemcpyAsync(s->planes + (size_t)src_r * s->plane_bytes, plane_host, s->plane_bytes, hipMemcpyHostToDevice, sl.st));
        if (s->stage) { // quarter from the full picture, sixteenth from the quarter, each with its replicated border (pic_analysis_process.c:2138-2200)
            const uint8_t* full = s->planes + (size_t)src_r * s->plane_bytes + (size_t)s->org_y * s->stride + s->org_x;
            uint8_t*       q    = s->lvl_planes[0] + (size_t)src_r * s->lvl_bytes[0];
            uint8_t*       x    = s->lvl_planes[1] + (size_t)src_r * s->lvl_bytes[1];
            svt_hip_downsample_2d_padded(full, s->stride, s->width, s->height, q, s->lvl_stride[0], s->lvl_pad[0], s->lvl_pad[0], 2, sl.st);
            svt_hip_downsample_2d_padded(q + (size_t)s->lvl_pad[0] * s->lvl_stride[0] + s->lvl_pad[0], s->lvl_stride[0], s->width >> 1, s->height >> 1, x,
                                         s->lvl_stride[1], s->lvl_pad[1], s->lvl_pad[1], 2, sl.st);
        }
        HIP_CHECK(hipEventRecord(s->uploaded[src_r], sl.st));
        s->ids[src_r] = pic_id;
    } else {
        HIP_CHECK(hipStreamWaitEvent(sl.st, s->uploaded[src_r], 0));
    }
    sl.reads.assign(ref_r.begin(), ref_r.end());
    sl.reads.push_back(src_r);
    if (n_refs == 0) { sl.busy = true; HIP_CHECK(hipEventRecord(sl.done, sl.st)); s->next_slot = (s->next_slot + 1) % (uint32_t)s->slots.size(); return si; }
    unsigned long long offs[64];
    for (uint32_t k = 0; k < n_refs && k < 64; k++) {
        HIP_CHECK(hipStreamWaitEvent(sl.st, s->uploaded[ref_r[k]], 0));
        offs[k] = (unsigned long long)ref_r[k] * s->plane_bytes;
    }
    const uint32_t n = s->sbs * n_refs;
    unsigned long long* d_offs = (unsigned long long*)(sl.descs + (size_t)s->sbs * s->max_refs);
    if (!stage) {
        HIP_CHECK(hipMemcpyAsync(d_offs, offs, n_refs * 8, hipMemcpyHostToDevice, sl.st));
        hipLaunchKernelGGL(me_build_descs_kernel, dim3((n + 255) / 256), dim3(256), 0, sl.st, sl.descs, (s->width + 63) / 64, s->sbs, n_refs, s->stride, s->org_x,
                           s->org_y, (unsigned long long)src_r * s->plane_bytes, (const unsigned long long*)d_offs, (int)area_w, (int)area_h);
        SVT_LAUNCH_CHECK();
        svt_hip_me_fullpel_search_batch(s->planes, s->planes, sl.descs, n, area_w, area_h, sub_sad, sl.sad, sl.mv, sl.ws, sl.st);
    } else { // HME levels 0-2 (one launch) -> final search centre + integer_search_b64 geometry + full-pel search
        const uint32_t sbs_x = (s->width + 63) / 64, sbs_y = s->sbs / sbs_x, aw = (s->width + 7) & ~7u, ah = (s->height + 7) & ~7u;
        const uint32_t regions = (uint32_t)stage->num_hme_sa_w * stage->num_hme_sa_h;
        const size_t   items = (size_t)n_refs * s->sbs * regions;
        SvtHipHmeLevelParams P[3];
        const uint8_t*       bases[3] = {s->lvl_planes[1], s->lvl_planes[0], s->planes};
        unsigned long long*  sads[3];
        int16_t*             scs[3];
        uint8_t* hp = sl.hme;
        for (int lv = 0; lv < 3; lv++) {
            sads[lv] = (unsigned long long*)hp; hp += svthip::align_up(s->hme_items * 8, 256);
            scs[lv]  = (int16_t*)hp;            hp += svthip::align_up(s->hme_items * 4, 256);
        }
        int16_t*            fin_sc  = (int16_t*)hp;            hp += svthip::align_up((size_t)s->max_refs * s->sbs * 4, 256);
        unsigned long long* fin_sad = (unsigned long long*)hp; hp += svthip::align_up((size_t)s->max_refs * s->sbs * 8, 256);
        void*               int_ws  = hp;
        for (int lv = 0; lv < 3; lv++) {
            SvtHipHmeLevelParams& L = P[lv];
            memset(&L, 0, sizeof(L));
            L.level = (uint8_t)lv; L.sub_sampled = stage->hme_sub_sampled; L.num_hme_sa_w = stage->num_hme_sa_w; L.num_hme_sa_h = stage->num_hme_sa_h;
            L.sa_width = stage->hme_sa_width[lv]; L.sa_height = stage->hme_sa_height[lv];
            L.sbs_x = sbs_x; L.sbs_y = sbs_y; L.n_refs = n_refs; L.prev_shift = lv == 1; L.aligned_width = aw; L.aligned_height = ah;
            const uint32_t stride = lv == 2 ? s->stride : s->lvl_stride[1 - lv], pad_x = lv == 2 ? s->org_x : s->lvl_pad[1 - lv],
                           pad_y = lv == 2 ? s->org_y : s->lvl_pad[1 - lv];
            const size_t   pb = lv == 2 ? s->plane_bytes : s->lvl_bytes[1 - lv];
            L.src_off = (uint64_t)src_r * pb + (uint64_t)pad_y * stride + pad_x;
            L.src_stride = stride; L.ref_stride = stride; L.ref_org_x = pad_x; L.ref_org_y = pad_y;
            L.ref_width = s->width >> (2 - lv); L.ref_height = s->height >> (2 - lv);
            for (uint32_t k = 0; k < n_refs; k++) L.ref_off[k] = (uint64_t)ref_r[k] * pb;
        }
        // init_me_hme_data leaves the centres at 0: one fill over the three levels' (sad, centre) arrays (the SADs are rewritten by the chain kernel)
        HIP_CHECK(hipMemsetAsync(sads[0], 0, (size_t)((uint8_t*)scs[2] - (uint8_t*)sads[0]) + items * 4, sl.st));
        // search_results[].do_ref of the stage lives where the formatting step expects it, so every pruning step carries over to me_prune_ref
        uint8_t* d_do_ref = nullptr;
        if (fmt) {
            const uint32_t  n_pus = fmt->enable_me_16x16 ? (fmt->enable_me_8x8 ? 85 : 21) : 5;
            const FmtLayout L     = fmt_layout(s->sbs, n_pus, fmt->max_refs, fmt->max_cand);
            d_do_ref = sl.fmt + L.do_ref;
            if (out->do_ref) HIP_CHECK(hipMemcpyAsync(d_do_ref, out->do_ref, (size_t)s->sbs * 8, hipMemcpyHostToDevice, sl.st));
            else HIP_CHECK(hipMemsetAsync(d_do_ref, 1, (size_t)s->sbs * 8, sl.st));
        }
        const uint8_t n_l0 = stage->results.num_of_ref_pic_to_search[0];
        const uint8_t list1_no_hme = !stage->me_type_mctf && !stage->temporal_layer_gt0 && n_refs > n_l0; // base layer, two lists (:1983, :2055, :2127, :2211)
        const int     hme_levels   = stage->hme_levels ? stage->hme_levels : 3;
        for (int lv = 0; lv < 3; lv++) {
            P[lv].n_refs_list0 = n_l0;
            for (uint32_t k = 0; k < n_refs; k++) P[lv].ref_pic_index[k] = stage->ref_pic_index[k];
        }
        uint32_t*           zz  = (uint32_t*)((uint8_t*)int_ws + s->int_ws);
        SvtHipPrehmeResult* pre = (SvtHipPrehmeResult*)((uint8_t*)zz + svthip::align_up((size_t)s->max_refs * s->sbs * 4, 256));
        const bool need_zz = stage->me_early_exit_th != 0 || stage->me_safe_limit_zz_th != 0; // hme_b64 only runs init_zz_sad then (:2444-2445), zz_sad_th alone has no effect
        if (need_zz) { // init_zz_sad: the zero-motion SAD gates pre-HME, HME levels 0 / 1, the integer search and (zz_sad_th) the reference list
            SvtHipMeIntegerSearchParams Z;
            memset(&Z, 0, sizeof(Z));
            Z.sbs_x = sbs_x; Z.sbs_y = sbs_y; Z.n_refs = n_refs; Z.aligned_width = aw; Z.aligned_height = ah;
            Z.src_off = (uint64_t)src_r * s->plane_bytes + (uint64_t)s->org_y * s->stride + s->org_x;
            Z.src_stride = s->stride; Z.ref_stride = s->stride; Z.ref_org_x = s->org_x; Z.ref_org_y = s->org_y;
            Z.n_refs_list0 = n_l0; Z.list1_no_hme = list1_no_hme;
            for (uint32_t k = 0; k < n_refs; k++) Z.ref_off[k] = (uint64_t)ref_r[k] * s->plane_bytes;
            svt_hip_me_zz_sad_batch(&Z, s->planes, s->planes, zz, sl.st);
            if (stage->zz_sad_th && d_do_ref) svt_hip_me_ref_gate_batch(&P[2], zz, stage->zz_sad_th, stage->zz_sad_pct, stage->temporal_layer_gt0, d_do_ref, sl.st);
            if (stage->me_safe_limit_zz_th && d_do_ref) svt_hip_me_ref_safe_limit_batch(&P[2], zz, stage->me_safe_limit_zz_th, d_do_ref, sl.st);
            if (stage->me_early_exit_th) P[0].zz_skip_th = P[1].zz_skip_th = stage->me_early_exit_th >> 2;
        }
        if (stage->prehme_enabled) {
            SvtHipPrehmeParams H;
            memset(&H, 0, sizeof(H));
            H.plane = P[0];
            for (int k = 0; k < 2; k++) {
                H.sa_min_width[k] = stage->prehme_sa_min_width[k]; H.sa_min_height[k] = stage->prehme_sa_min_height[k];
                H.sa_max_width[k] = stage->prehme_sa_max_width[k]; H.sa_max_height[k] = stage->prehme_sa_max_height[k];
            }
            for (uint32_t k = 0; k < n_refs; k++) // svt_aom_get_scaled_picture_distance (:1239-1243) when the caller's distances are the raw ones
                H.hme_sr_factor[k] = stage->me_type_mctf ? (uint16_t)(stage->dist[k] * 5 / 8 + (stage->dist[k] % 8 ? 1 : 0)) : stage->dist[k];
            H.skip_search_line = stage->prehme_skip_search_line; H.l1_early_exit = stage->prehme_l1_early_exit; H.temporal_layer_gt0 = stage->temporal_layer_gt0;
            H.me_early_exit_th = stage->me_early_exit_th; H.phme_sad_th = stage->phme_sad_th; H.phme_sad_pct = stage->phme_sad_pct;
            svt_hip_prehme_batch(&H, s->lvl_planes[1], s->lvl_planes[1], need_zz ? zz : nullptr, d_do_ref, pre, sl.st);
            P[0].prehme_enabled = 1;
        }
        if (stage->hme_l0_per_ref) {
            P[0].per_ref_area = 1;
            for (uint32_t k = 0; k < n_refs; k++) { P[0].sa_width_ref[k] = stage->hme_l0_sa_width_ref[k]; P[0].sa_height_ref[k] = stage->hme_l0_sa_height_ref[k]; }
            if (stage->reduce_hme_l0_sr_th_min && stage->reduce_hme_l0_sr_th_max) { // level-0 areas resized from list 0 / reference 0's level-0 motion (low-delay settings)
                P[0].l0_mv_th_min = stage->reduce_hme_l0_sr_th_min; P[0].l0_mv_th_max = stage->reduce_hme_l0_sr_th_max;
                for (uint32_t k = 0; k < n_refs; k++) { P[0].sa_width_ref2[k] = stage->hme_l0_sa_width_ref2[k]; P[0].sa_height_ref2[k] = stage->hme_l0_sa_height_ref2[k]; }
                if (stage->sr_adjustment == 2) {
                    P[0].l0_still_rule = 1;
                    for (uint32_t k = 0; k < n_refs; k++) { P[0].sa_width_ref4[k] = stage->hme_l0_sa_width_ref4[k]; P[0].sa_height_ref4[k] = stage->hme_l0_sa_height_ref4[k]; }
                }
            }
        }
        SvtHipHmeChainInputs in;
        memset(&in, 0, sizeof(in));
        in.prev_me_stage_based_exit_th = stage->prev_me_stage_based_exit_th;
        in.zz_sad = stage->me_early_exit_th ? zz : nullptr; in.do_ref = d_do_ref; in.prehme = stage->prehme_enabled ? pre : nullptr;
        in.n_levels = (uint8_t)hme_levels; in.list1_no_hme = list1_no_hme;
        svt_hip_hme_chain_batch(P, bases, bases, &in, (uint64_t* const*)sads, scs, sl.st);
        SvtHipMeIntegerSearchParams Q;
        memset(&Q, 0, sizeof(Q));
        Q.sbs_x = sbs_x; Q.sbs_y = sbs_y; Q.n_refs = n_refs; Q.regions = regions; Q.aligned_width = aw; Q.aligned_height = ah;
        Q.sa_min_width = stage->me_sa_min_width; Q.sa_min_height = stage->me_sa_min_height;
        Q.sa_max_width = stage->me_sa_max_width; Q.sa_max_height = stage->me_sa_max_height;
        Q.sub_sad = stage->me_sub_sad; Q.mv_adj_enabled = stage->mv_adj_enabled; Q.mv_adj_nearest_ref_only = stage->mv_adj_nearest_ref_only;
        Q.mv_adj_mv_size_th = stage->mv_adj_mv_size_th; Q.mv_adj_sa_multiplier = stage->mv_adj_sa_multiplier;
        for (uint32_t k = 0; k < n_refs; k++) { Q.dist[k] = stage->dist[k]; Q.ref_pic_index[k] = stage->ref_pic_index[k]; Q.ref_off[k] = (uint64_t)ref_r[k] * s->plane_bytes; }
        Q.src_off = (uint64_t)src_r * s->plane_bytes + (uint64_t)s->org_y * s->stride + s->org_x;
        Q.src_stride = s->stride; Q.ref_stride = s->stride; Q.ref_org_x = s->org_x; Q.ref_org_y = s->org_y;
        Q.tf_me_exit_th = stage->me_type_mctf ? stage->tf_me_exit_th : 0;
        Q.list1_no_hme = list1_no_hme;
        Q.n_refs_list0 = stage->results.num_of_ref_pic_to_search[0];
        Q.hme_prune_enabled = stage->hme_prune_enabled; Q.prune_ref_if_hme_sad_dev_bigger_than_th = stage->prune_ref_if_hme_sad_dev_bigger_than_th;
        Q.sr_adjustment = stage->sr_adjustment; Q.reduce_me_sr_based_on_mv_length_th = stage->reduce_me_sr_based_on_mv_length_th;
        Q.stationary_hme_sad_abs_th = stage->stationary_hme_sad_abs_th; Q.stationary_me_sr_divisor = stage->stationary_me_sr_divisor;
        Q.reduce_me_sr_based_on_hme_sad_abs_th = stage->reduce_me_sr_based_on_hme_sad_abs_th;
        Q.me_sr_divisor_for_low_hme_sad = stage->me_sr_divisor_for_low_hme_sad; Q.me_early_exit_th = stage->me_early_exit_th;
        Q.is_ref = stage->is_ref; Q.me_8x8_var_enabled = stage->me_8x8_var_enabled; Q.me_sr_div4_th = stage->me_sr_div4_th;
        Q.me_sr_div2_th = stage->me_sr_div2_th; Q.me_sr_mult2_th = stage->me_sr_mult2_th; Q.ref_width = s->width; Q.ref_height = s->height;
        svt_hip_me_integer_search_batch(&Q, s->planes, s->planes, (const uint64_t*)sads[hme_levels - 1], scs[hme_levels - 1], d_do_ref, nullptr, stage->me_early_exit_th ? zz : nullptr, sl.sad, sl.mv, fin_sc,
                                        (uint64_t*)fin_sad, int_ws, sl.st);
        if (out && out->hme_sc) HIP_CHECK(hipMemcpyAsync(out->hme_sc, fin_sc, (size_t)n * 4, hipMemcpyDeviceToHost, sl.st));
        if (out && out->hme_sad) HIP_CHECK(hipMemcpyAsync(out->hme_sad, fin_sad, (size_t)n * 8, hipMemcpyDeviceToHost, sl.st));
    }
    if (best_sad_host) HIP_CHECK(hipMemcpyAsync(best_sad_host, sl.sad, (size_t)n * SVT_HIP_ME_NUM_BLOCKS * 4, hipMemcpyDeviceToHost, sl.st));
    if (best_mv_host) HIP_CHECK(hipMemcpyAsync(best_mv_host, sl.mv, (size_t)n * SVT_HIP_ME_NUM_BLOCKS * 4, hipMemcpyDeviceToHost, sl.st));
    if (fmt) { // the stage's final product: MeSbResults + per-SB statistics, formatted on the device from the tables just written
        SvtHipMeResultsParams P = *fmt;
        P.n_sb = s->sbs;
        const uint32_t  n_pus = P.enable_me_16x16 ? (P.enable_me_8x8 ? 85 : 21) : 5;
        const FmtLayout L     = fmt_layout(s->sbs, n_pus, P.max_refs, P.max_cand);
        HIP_CHECK(hipMemsetAsync(sl.fmt + L.total, 0, L.stats - L.total, sl.st)); // entries the reference leaves unwritten read as 0
        if (!stage) { // (the stage form initialised do_ref before the integer search, which may have pruned references)
            if (out->do_ref) HIP_CHECK(hipMemcpyAsync(sl.fmt + L.do_ref, out->do_ref, (size_t)s->sbs * 8, hipMemcpyHostToDevice, sl.st));
            else HIP_CHECK(hipMemsetAsync(sl.fmt + L.do_ref, 1, (size_t)s->sbs * 8, sl.st));
        }
        svt_hip_me_results_batch(&P, sl.sad, sl.mv, sl.fmt + L.do_ref, s->sb_size, sl.fmt + L.total, (uint32_t*)(sl.fmt + L.mv), sl.fmt + L.cand,
                                 (SvtHipMeSbStats*)(sl.fmt + L.stats), sl.st);
        if (out->do_ref) HIP_CHECK(hipMemcpyAsync(out->do_ref, sl.fmt + L.do_ref, (size_t)s->sbs * 8, hipMemcpyDeviceToHost, sl.st));
        HIP_CHECK(hipMemcpyAsync(out->total_me_candidate_index, sl.fmt + L.total, (size_t)s->sbs * n_pus, hipMemcpyDeviceToHost, sl.st));
        HIP_CHECK(hipMemcpyAsync(out->me_candidate_array, sl.fmt + L.cand, (size_t)s->sbs * n_pus * P.max_cand, hipMemcpyDeviceToHost, sl.st));
        HIP_CHECK(hipMemcpyAsync(out->me_mv_array, sl.fmt + L.mv, (size_t)s->sbs * n_pus * P.max_refs * 4, hipMemcpyDeviceToHost, sl.st));
        HIP_CHECK(hipMemcpyAsync(out->sb_stats, sl.fmt + L.stats, (size_t)s->sbs * sizeof(SvtHipMeSbStats), hipMemcpyDeviceToHost, sl.st));
    }
    HIP_CHECK(hipEventRecord(sl.done, sl.st));
    sl.busy      = true;
    s->next_slot = (s->next_slot + 1) % (uint32_t)s->slots.size();
    return si;
}

int svt_hip_me_session_submit(void* session, int64_t pic_id, const uint8_t* plane_host, const int64_t* ref_ids, uint32_t n_refs, uint32_t area_w,
                              uint32_t area_h, int sub_sad, uint32_t* best_sad_host, uint32_t* best_mv_host) {
    SVT_HIP_ENTRY_TRY
    return me_session_submit(session, pic_id, plane_host, ref_ids, n_refs, area_w, area_h, sub_sad, best_sad_host, best_mv_host, nullptr, nullptr);
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}
int svt_hip_me_session_submit_results(void* session, int64_t pic_id, const uint8_t* plane_host, const int64_t* ref_ids, uint32_t n_refs, uint32_t area_w,
                                      uint32_t area_h, int sub_sad, const SvtHipMeResultsParams* params, const SvtHipMeResultsHost* out) {
    SVT_HIP_ENTRY_TRY
    return me_session_submit(session, pic_id, plane_host, ref_ids, n_refs, area_w, area_h, sub_sad, out->best_sad, out->best_mv, params, out);
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}

int svt_hip_me_session_enable_stage(void* session, uint32_t quarter_pad, uint32_t sixteenth_pad, uint32_t max_regions, uint32_t max_me_area_width,
                                    uint32_t max_me_area_height) {
    SVT_HIP_ENTRY_TRY
    Session* s = (Session*)session;
    svthip::DeviceGuard guard(s->device);
    if (s->stage || !max_regions || s->max_refs > 8) return -1;
    const uint32_t pads[2] = {quarter_pad, sixteenth_pad};
    for (int k = 0; k < 2; k++) {
        const uint32_t w = s->width >> (k + 1), h = s->height >> (k + 1);
        s->lvl_pad[k]    = pads[k];
        s->lvl_stride[k] = w + 2 * pads[k];
        s->lvl_rows[k]   = h + 2 * pads[k] + (64u >> (k + 1)); // the last SB row overhangs the picture
        s->lvl_bytes[k]  = (size_t)s->lvl_stride[k] * s->lvl_rows[k];
        HIP_CHECK(hipMalloc((void**)&s->lvl_planes[k], s->lvl_bytes[k] * s->ring));
        HIP_CHECK(hipMemset(s->lvl_planes[k], 0, s->lvl_bytes[k] * s->ring));
    }
    s->max_regions = max_regions; s->max_area_w = max_me_area_width; s->max_area_h = max_me_area_height;
    s->hme_items = (size_t)s->max_refs * s->sbs * max_regions;
    const size_t n = (size_t)s->sbs * s->max_refs;
    { // workspace of the integer-search stage at the largest area the caller announced (including the variance probe's 3/2 enlargement)
        SvtHipMeIntegerSearchParams D;
        memset(&D, 0, sizeof(D));
        D.sbs_x = (s->width + 63) / 64; D.sbs_y = s->sbs / D.sbs_x; D.n_refs = s->max_refs; D.regions = 1;
        D.sa_min_width = D.sa_max_width = (int16_t)max_me_area_width; D.sa_min_height = D.sa_max_height = (int16_t)max_me_area_height;
        for (int k = 0; k < 8; k++) D.dist[k] = 1;
        D.me_8x8_var_enabled = 1;
        s->int_ws = svt_hip_me_integer_search_workspace(&D);
    }
    const size_t per_slot = 3 * (svthip::align_up(s->hme_items * 8, 256) + svthip::align_up(s->hme_items * 4, 256)) + svthip::align_up(n * 4, 256) +
                            svthip::align_up(n * 8, 256) + s->int_ws + svthip::align_up(n * 4, 256) + svthip::align_up(n * 2 * sizeof(SvtHipPrehmeResult), 256) +
                            256; // ... + zz_sad + pre-HME results
    for (auto& sl : s->slots) HIP_CHECK(hipMalloc((void**)&sl.hme, per_slot));
    s->stage = true;
    return 0;
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}
int svt_hip_me_session_submit_stage(void* session, int64_t pic_id, const uint8_t* plane_host, const int64_t* ref_ids, uint32_t n_refs,
                                    const SvtHipMeStageParams* stage, const SvtHipMeResultsHost* out) {
    SVT_HIP_ENTRY_TRY
    if (!stage) return -5;
    const bool fmt = n_refs > 0 && out && out->total_me_candidate_index;
    return me_session_submit(session, pic_id, plane_host, ref_ids, n_refs, 0, 0, 0, out ? out->best_sad : nullptr, out ? out->best_mv : nullptr,
                             fmt ? &stage->results : nullptr, out, stage);
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}

// Forget a resident picture (its host content changed, e.g. after in-place temporal filtering): the next submission that names it as the source uploads it again.
void svt_hip_me_session_invalidate(void* session, int64_t pic_id) {
    SVT_HIP_ENTRY_TRY
    Session* s = (Session*)session;
    if (!s) return;
    for (uint32_t r = 0; r < s->ring; r++)
        if (s->ids[r] == pic_id) s->ids[r] = -1;
    SVT_HIP_ENTRY_CATCH((void)0)
}
// 1 when the picture is resident in the ring (usable as a reference), else 0
int svt_hip_me_session_resident(void* session, int64_t pic_id) {
    SVT_HIP_ENTRY_TRY
    Session* s = (Session*)session;
    if (!s) return 0;
    for (uint32_t r = 0; r < s->ring; r++)
        if (s->ids[r] == pic_id) return 1;
    return 0;
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}

int svt_hip_me_session_wait(void* session, int slot) {
    SVT_HIP_ENTRY_TRY
    Session* s = (Session*)session;
    svthip::DeviceGuard guard(s->device);
    if (slot < 0 || slot >= (int)s->slots.size()) return 0;
    HIP_CHECK(hipEventSynchronize(s->slots[slot].done));
    return 0;
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}

} // extern "C"

SVT_HIP_DEFINE_WARM(me_session) // (svt_hip_warmup loads this translation unit's code object at encoder initialisation: svt_hip_common.h)
