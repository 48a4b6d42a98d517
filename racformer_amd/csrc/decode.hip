// decode.hip -- NMS-free decode of one sample as ONE kernel (gfx950): sigmoid -> top-K of the Q x C scores ->
// label / query index -> denormalize_bbox -> centre-range and score masks -> z at the box bottom.
// Replaces NMSFreeCoder.decode_single + the box reshuffle of get_bboxes (models/bbox/coders/nms_free_coder.py:37-88,
// models/bbox/utils.py:26-46, models/racformer_head.py:488-507) -- in torch ~15 launches (sigmoid, topk's two kernels,
// sort, gathers, exp / atan2, comparisons, concatenations).  Shape-static output [K][11] = (x, y, z_bottom, w, l, h, yaw,
// vx, vy, score, label); rows that fail the masks carry score = -1 (the data-parallel wire format).
//
// One workgroup of 1024 threads.  Top-K by radix select on 48-bit keys (order-preserving bits of the logit, then the
// inverted flat index: all keys distinct, ties resolve to the smaller index, the result is deterministic), 8 bits per pass
// on a 256-bin LDS histogram whose suffix sums are scanned in parallel; the K survivors are ordered by rank counting.  sigmoid is monotonic, so
// selection runs on the logits and only K sigmoids are evaluated.
#include "rac_common.h"

#define DEC_THREADS 1024
#define DEC_MAX_PER_THREAD 16   /* Q*C <= 16384 */
#define DEC_MAX_K 512

struct DecArgs {
    const float *cls;   // [Q][C] logits
    const float *box;   // [Q][10]: cx, cy, log w, log l, cz, log h, sin, cos, vx, vy
    float *out;         // [K][11]
    float range[6];
    float thr;
    int Q, C, K, use_thr;
};

__device__ __forceinline__ unsigned dec_sortable(float v)
{
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // larger float <=> larger unsigned
}

__global__ __launch_bounds__(DEC_THREADS) void decode_topk_kernel(const DecArgs a)
{
    __shared__ unsigned hist[256], scan[256];
    __shared__ unsigned long long sel[DEC_MAX_K];
    __shared__ unsigned long long s_prefix, s_mask;
    __shared__ int s_remaining, s_count;
    const int tid = threadIdx.x;
    const int n = a.Q * a.C;

    unsigned long long key[DEC_MAX_PER_THREAD];
#pragma unroll
    for (int j = 0; j < DEC_MAX_PER_THREAD; ++j) {
        const int i = tid + DEC_THREADS * j;
        key[j] = 0ull;   // below every real key (real keys have a non-zero high word unless the logit is -NaN-like)
        if (i < n)
            key[j] = ((unsigned long long)dec_sortable(a.cls[i]) << 16) | (unsigned long long)(0xFFFFu - (unsigned)i);
    }
    if (tid == 0) {
        s_prefix = 0ull;
        s_mask = 0ull;
        s_remaining = a.K;
        s_count = 0;
    }
    __syncthreads();
    // radix select of the K-th largest key: 6 passes of 8 bits, most significant first
    for (int shift = 40; shift >= 0; shift -= 8) {
        if (tid < 256)
            hist[tid] = 0u;
        __syncthreads();
        const unsigned long long prefix = s_prefix, mask = s_mask;
        const int remaining = s_remaining;
#pragma unroll
        for (int j = 0; j < DEC_MAX_PER_THREAD; ++j)
            if (tid + DEC_THREADS * j < n && (key[j] & mask) == prefix)
                atomicAdd(&hist[(unsigned)(key[j] >> shift) & 255u], 1u);
        __syncthreads();
        // suffix sums S[b] = sum_{c >= b} hist[c] (Hillis-Steele over 256 bins, threads 0..255); the selected bin is the
        // largest b with S[b] >= remaining, i.e. S[b] >= remaining > S[b+1]
        if (tid < 256)
            scan[tid] = hist[tid];
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const unsigned add = (tid < 256 && tid + off < 256) ? scan[tid + off] : 0u;
            __syncthreads();
            if (tid < 256)
                scan[tid] += add;
            __syncthreads();
        }
        if (tid < 256) {
            const unsigned sb = scan[tid], sn = tid + 1 < 256 ? scan[tid + 1] : 0u;
            if ((sb >= (unsigned)remaining && sn < (unsigned)remaining) || (tid == 0 && sb < (unsigned)remaining)) {
                s_remaining = remaining - (int)sn;      // (tid 0 with too few candidates: everything is selected)
                s_prefix = prefix | ((unsigned long long)tid << shift);
                s_mask = mask | (0xFFull << shift);
            }
        }
        __syncthreads();
    }
    const unsigned long long kth = s_prefix;   // exactly K keys are >= kth (keys are distinct)
    if (tid < DEC_MAX_K)
        sel[tid] = 0ull;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < DEC_MAX_PER_THREAD; ++j)
        if (tid + DEC_THREADS * j < n && key[j] >= kth) {
            const int pos = atomicAdd(&s_count, 1);
            if (pos < DEC_MAX_K)
                sel[pos] = key[j];
        }
    __syncthreads();
    // order by rank counting: the rank of a survivor is the number of survivors with a larger key (keys are distinct)
    const int nsel = min(s_count, DEC_MAX_K);
    unsigned long long mykey = 0ull;
    int rank = -1;
    if (tid < nsel) {
        mykey = sel[tid];
        rank = 0;
        for (int i = 0; i < nsel; ++i)
            rank += sel[i] > mykey ? 1 : 0;
    }
    __syncthreads();
    if (tid < DEC_MAX_K)
        sel[tid] = 0ull;
    __syncthreads();
    if (rank >= 0)
        sel[rank] = mykey;
    __syncthreads();
    if (tid < a.K) {
        const unsigned long long kk = sel[tid];
        const int idx = (int)(0xFFFFu - (unsigned)(kk & 0xFFFFull));
        float *o = a.out + (size_t)tid * 11;
        if (kk == 0ull || idx < 0 || idx >= n) {   // fewer than K candidates (K > Q*C): empty row
#pragma unroll
            for (int c = 0; c < 11; ++c)
                o[c] = c == 9 ? -1.f : 0.f;
            return;
        }
        const int q = idx / a.C, label = idx - q * a.C;
        const float score = 1.f / (1.f + expf(-a.cls[idx]));
        const float *bx = a.box + (size_t)q * 10;
        const float cx = bx[0], cy = bx[1], cz = bx[4];
        const float w = expf(bx[2]), l = expf(bx[3]), h = expf(bx[5]);
        const float yaw = atan2f(bx[6], bx[7]);
        bool keep = cx >= a.range[0] && cy >= a.range[1] && cz >= a.range[2] && cx <= a.range[3] && cy <= a.range[4] && cz <= a.range[5];
        if (a.use_thr)
            keep = keep && score > a.thr;
        o[0] = cx; o[1] = cy; o[2] = cz - h * 0.5f; o[3] = w; o[4] = l; o[5] = h; o[6] = yaw; o[7] = bx[8]; o[8] = bx[9];
        o[9] = keep ? score : -1.f;
        o[10] = (float)label;
    }
}

extern "C" int rac_decode_fwd(const float *cls_scores, const float *bbox_preds, float *out, int num_query, int num_classes,
                              int max_num, const float *post_center_range, float score_threshold, int use_threshold,
                              void *stream)
{
    RAC_CHECK_ARG(num_query >= 1 && num_classes >= 1 && (long)num_query * num_classes <= (long)DEC_THREADS * DEC_MAX_PER_THREAD,
                  "rac_decode_fwd: Q*C = %ld exceeds %d", (long)num_query * num_classes, DEC_THREADS * DEC_MAX_PER_THREAD);
    RAC_CHECK_ARG(max_num >= 1 && max_num <= DEC_MAX_K, "rac_decode_fwd: max_num=%d (1..%d)", max_num, DEC_MAX_K);
    RAC_CHECK_ARG(cls_scores && bbox_preds && out && post_center_range, "rac_decode_fwd: null pointer");
    DecArgs a;
    a.cls = cls_scores; a.box = bbox_preds; a.out = out;
    for (int i = 0; i < 6; ++i)
        a.range[i] = post_center_range[i];
    a.thr = score_threshold; a.use_thr = use_threshold;
    a.Q = num_query; a.C = num_classes; a.K = max_num;
    hipLaunchKernelGGL(decode_topk_kernel, dim3(1), dim3(DEC_THREADS), 0, (hipStream_t)stream, a);
    return rac_launch_status("rac_decode_fwd");
}
