// sasa.hip -- scale-adaptive self-attention core as one kernel (gfx950).
//
// Replaces, per decoder layer, calc_bbox_dists (a [B,Q,Q] cdist), the [B*8,Q,Q] float mask, and
// nn.MultiheadAttention's QK^T / softmax / AV (models/racformer_transformer.py:296-335 over mmcv's
// MultiheadAttention wrapper): nothing of size Q x Q ever touches HBM.
//   logits[b,h,i,j] = (q_i / sqrt(d)) . k_j  -  ||c_i - c_j||_2 * tau[b,i,h]
//   out[b,i,h*d:(h+1)*d] = softmax_j(logits) @ v
// with c = box centres in metres (decode_bbox(theta_d2xy(query_bbox))[:2], :301-302,:325).
// in_proj / out_proj stay library GEMMs outside.
//
// Mapping: a workgroup = 16 query rows of one (batch, head); a 16-lane group owns a row.  K/V are
// streamed through LDS in 64-key tiles (next tile prefetched into registers while the current one
// is consumed; row stride 36 floats = conflict-free ds_read_b128); lane r of a row scores keys
// r, r+16, r+32, r+48 of the tile and folds them into its own online-softmax state (running max,
// sum, d-wide accumulator) with one rescale per 4 keys; the 16 partial states of a row are merged
// with a log-sum-exp butterfly over the 16 lanes.  Centres of all keys are computed once per
// workgroup into LDS.  fp32 throughout (exact-fp32 VALU; fp32 MFMA has the same rate on gfx950).
#include "rac_common.h"

#define SASA_D 32
#define SASA_ROWS 16
#define SASA_TWO_PI 6.283185307179586f

struct SasaArgs {
    const float *qkv;   // [B,Q,3,H,d]  (in_proj output)
    const float *tau;   // [B,Q,H], row stride ld_tau
    const float *qbox;  // [B,Q,10]
    const float *box;   // optional [B,Q,8] from rac_box_prep_fwd (cx, cy, ...): skips the trig prologue
    float *out;         // [B,Q,H*d]
    float pc[6];
    int B, Q, H, ld_tau, ld_qkv;
    int row_blocks;
};

#define SASA_TILE 64   /* keys per LDS tile */
#define SASA_KS 36     /* LDS row stride (floats): 16-byte aligned and conflict-free for ds_read_b128 */

__global__ __launch_bounds__(256) void sasa_d32_kernel(const SasaArgs a)
{
    extern __shared__ float smem[];
    float *scen = smem;                       // [Q][2] key centres (metres)
    float *sK = smem + 2 * ((a.Q + 1) & ~1);  // [64][36]
    float *sV = sK + SASA_TILE * SASA_KS;     // [64][36]
    const int tid = threadIdx.x;
    const int r = tid & 15, row = tid >> 4;
    int bid = blockIdx.x;
    const int rb = bid % a.row_blocks; bid /= a.row_blocks;
    const int h = bid % a.H;
    const int b = bid / a.H;
    const int Q = a.Q, H = a.H;
    const size_t tok = (size_t)b * Q;
    const int ld = a.ld_qkv;  // floats per token row (>= 3*H*d: qkv may be a column slice of a wider GEMM output)

    // K/V tile staging: thread -> (key = tid/4 (0..63), 2 float4 of K and 2 of V at columns (tid%4)*2..)
    const int sk = tid >> 2, sc = (tid & 3) * 2;
    rac_f4 pk[2], pv[2];
    auto prefetch = [&](int tile) {
        const int j = tile * SASA_TILE + sk;
        const int jj = j < Q ? j : Q - 1;
        const rac_f4 *kp = reinterpret_cast<const rac_f4 *>(a.qkv + (tok + jj) * ld + (H + h) * SASA_D);
        const rac_f4 *vp = reinterpret_cast<const rac_f4 *>(a.qkv + (tok + jj) * ld + (2 * H + h) * SASA_D);
        pk[0] = kp[sc]; pk[1] = kp[sc + 1];
        pv[0] = vp[sc]; pv[1] = vp[sc + 1];
    };
    prefetch(0);

    for (int j = tid; j < Q; j += 256) {
        const float *qb = a.qbox + ((size_t)b * Q + j) * 10;
        const float ang = qb[0] * SASA_TWO_PI, rad = qb[1] * 65.0f;
        const float xn = fminf(fmaxf((51.2f + rad * cosf(ang)) / 102.4f, 0.f), 1.f);
        const float yn = fminf(fmaxf((51.2f + rad * sinf(ang)) / 102.4f, 0.f), 1.f);
        scen[2 * j] = xn * (a.pc[3] - a.pc[0]) + a.pc[0];
        scen[2 * j + 1] = yn * (a.pc[4] - a.pc[1]) + a.pc[1];
    }

    const int i = rb * SASA_ROWS + row;
    const bool live = i < Q;
    const int ii = live ? i : Q - 1;
    const float scale = 0.17677669529663687f;  // sqrt(1/32) as torch computes math.sqrt(1.0/d)
    float q[SASA_D];
    {
        const rac_f4 *qp = reinterpret_cast<const rac_f4 *>(a.qkv + (tok + ii) * ld + h * SASA_D);
#pragma unroll
        for (int c = 0; c < SASA_D / 4; ++c) {
            const rac_f4 t = qp[c];
            q[4 * c] = t.x * scale; q[4 * c + 1] = t.y * scale; q[4 * c + 2] = t.z * scale; q[4 * c + 3] = t.w * scale;
        }
    }
    const float tau = a.tau[(tok + ii) * a.ld_tau + h];
    __syncthreads();  // centres visible
    const float cix = scen[2 * ii], ciy = scen[2 * ii + 1];

    float m = -INFINITY, l = 0.f;
    float acc[SASA_D];
#pragma unroll
    for (int c = 0; c < SASA_D; ++c)
        acc[c] = 0.f;

    const int ntiles = (Q + SASA_TILE - 1) / SASA_TILE;
    for (int tile = 0; tile < ntiles; ++tile) {
        // publish the prefetched tile, start fetching the next one
        *reinterpret_cast<rac_f4 *>(sK + sk * SASA_KS + sc * 4) = pk[0];
        *reinterpret_cast<rac_f4 *>(sK + sk * SASA_KS + sc * 4 + 4) = pk[1];
        *reinterpret_cast<rac_f4 *>(sV + sk * SASA_KS + sc * 4) = pv[0];
        *reinterpret_cast<rac_f4 *>(sV + sk * SASA_KS + sc * 4 + 4) = pv[1];
        __syncthreads();
        if (tile + 1 < ntiles)
            prefetch(tile + 1);
        // lane r scores keys r, r+16, r+32, r+48 of the tile
        float sc4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kl = r + 16 * u, j = tile * SASA_TILE + kl;
            const rac_f4 *kp = reinterpret_cast<const rac_f4 *>(sK + kl * SASA_KS);
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < SASA_D / 4; ++c) {
                const rac_f4 kk = kp[c];
                s += q[4 * c] * kk.x + q[4 * c + 1] * kk.y + q[4 * c + 2] * kk.z + q[4 * c + 3] * kk.w;
            }
            const int jc = j < Q ? j : Q - 1;
            const float dx = cix - scen[2 * jc], dy = ciy - scen[2 * jc + 1];
            s += -sqrtf(dx * dx + dy * dy) * tau;
            sc4[u] = j < Q ? s : -INFINITY;
        }
        const float mn = fmaxf(fmaxf(m, fmaxf(sc4[0], sc4[1])), fmaxf(sc4[2], sc4[3]));
        if (mn > -INFINITY) {
            const float corr = (m == -INFINITY) ? 0.f : expf(m - mn);
            l *= corr;
#pragma unroll
            for (int c = 0; c < SASA_D; ++c)
                acc[c] *= corr;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float p = (sc4[u] == -INFINITY) ? 0.f : expf(sc4[u] - mn);
                l += p;
                const rac_f4 *vp = reinterpret_cast<const rac_f4 *>(sV + (r + 16 * u) * SASA_KS);
#pragma unroll
                for (int c = 0; c < SASA_D / 4; ++c) {
                    const rac_f4 vv = vp[c];
                    acc[4 * c] += p * vv.x; acc[4 * c + 1] += p * vv.y; acc[4 * c + 2] += p * vv.z; acc[4 * c + 3] += p * vv.w;
                }
            }
            m = mn;
        }
        __syncthreads();  // everyone done with this tile before it is overwritten
    }
    // merge the 16 lanes of the row: log-sum-exp butterfly
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
        const float mo = __shfl_xor(m, off, 16);
        const float lo = __shfl_xor(l, off, 16);
        const float mn = fmaxf(m, mo);
        const float ca = (m == -INFINITY) ? 0.f : expf(m - mn);
        const float cb = (mo == -INFINITY) ? 0.f : expf(mo - mn);
        l = l * ca + lo * cb;
#pragma unroll
        for (int c = 0; c < SASA_D; ++c) {
            const float ao = __shfl_xor(acc[c], off, 16);
            acc[c] = acc[c] * ca + ao * cb;
        }
        m = mn;
    }
    if (live) {
        // lane r writes channels 2r, 2r+1 (all lanes hold the merged state)
        const float inv = 1.f / l;
        float o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int c = 0; c < SASA_D / 2; ++c)
            if (c == r) {
                o0 = acc[2 * c];
                o1 = acc[2 * c + 1];
            }
        float2 o = make_float2(o0 * inv, o1 * inv);
        *reinterpret_cast<float2 *>(a.out + (tok + i) * (size_t)(H * SASA_D) + h * SASA_D + 2 * r) = o;
    }
}

// ---------------------------------------------------------------------------------------------------
// Matrix-core version (default): QK^T and PV on v_mfma_f32_16x16x4_f32 (exact fp32).
// Workgroup = 16 queries of one (batch, head); its 4 waves take the 16-key tiles round-robin.
//   S^T tile [16 keys x 16 queries] = K_tile (A operand) . Q^T (B operand): 8 MFMAs.  With the k index of
//   a group of four steps assigned as k = 16u + 4*lk + i (as in rowgemm.hip) every operand is a 16-byte load.
//   The accumulator then holds S^T[key = 4*lk + r][query = li] -- exactly the B-operand layout of the
//   second product O^T[32 ch x 16 queries] = V^T (A operand) . P^T (B operand), so the probabilities go
//   from the first product's accumulators into the second product's operands without leaving registers.
//   Softmax statistics are per query = per lane column: a register reduction, two DPP steps over lk and
//   one LDS exchange between the four waves.
typedef float sasa_f4 __attribute__((ext_vector_type(4)));
#define SASA_NT 16 /* key tiles per wave: Q <= 4*16*16 = 1024 */

__global__ __launch_bounds__(256) void sasa_mfma_kernel(const SasaArgs a)
{
    extern __shared__ float smem[];
    float *scen = smem;                       // [Q][2] key centres (metres)
    float *sred = smem + 2 * ((a.Q + 1) & ~1);  // [4 waves][16] max, then [4][16] sum
    float *so = sred + 2 * 4 * 16;            // [4 waves][32 ch][16 queries]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    int bid = blockIdx.x;
    const int rb = bid % a.row_blocks; bid /= a.row_blocks;
    const int h = bid % a.H;
    const int b = bid / a.H;
    const int Q = a.Q, H = a.H;
    const size_t tok = (size_t)b * Q;
    const int ld = a.ld_qkv;

    for (int j = tid; j < Q; j += 256) {
        if (a.box) {
            scen[2 * j] = a.box[((size_t)b * Q + j) * 8];
            scen[2 * j + 1] = a.box[((size_t)b * Q + j) * 8 + 1];
        } else {
            const float *qb = a.qbox + ((size_t)b * Q + j) * 10;
            const float ang = qb[0] * SASA_TWO_PI, rad = qb[1] * 65.0f;
            const float xn = fminf(fmaxf((51.2f + rad * cosf(ang)) / 102.4f, 0.f), 1.f);
            const float yn = fminf(fmaxf((51.2f + rad * sinf(ang)) / 102.4f, 0.f), 1.f);
            scen[2 * j] = xn * (a.pc[3] - a.pc[0]) + a.pc[0];
            scen[2 * j + 1] = yn * (a.pc[4] - a.pc[1]) + a.pc[1];
        }
    }
    // this lane's query column
    const int qi = rb * SASA_ROWS + li;
    const int qc = qi < Q ? qi : Q - 1;
    const float scale = 0.17677669529663687f;
    rac_f4 qb4[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        qb4[u] = rac_ld4(a.qkv + (tok + qc) * ld + h * SASA_D + 16 * u + 4 * lk);
        qb4[u].x *= scale; qb4[u].y *= scale; qb4[u].z *= scale; qb4[u].w *= scale;
    }
    const float tau = a.tau[(tok + qc) * a.ld_tau + h];
    __syncthreads();
    const float cqx = scen[2 * qc], cqy = scen[2 * qc + 1];

    const int ntiles = (Q + 15) >> 4;
    sasa_f4 sc[SASA_NT];
    float mloc = -INFINITY;
    // K rows one tile ahead of their MFMAs (two register sets; the loads of tile jt + 1 are pinned above the MFMAs of tile
    // jt): without this every tile waited out its own L2 latency -- 15 round trips per wave
    auto load_k = [&](int jt, rac_f4 &k0, rac_f4 &k1) {
        const int tile = wave + 4 * jt;
        const int key = tile * 16 + li;
        const int kc = key < Q ? key : Q - 1;
        const float *kp = a.qkv + (tok + kc) * ld + (H + h) * SASA_D + 4 * lk;
        k0 = rac_ld4(kp);
        k1 = rac_ld4(kp + 16);
    };
    rac_f4 kb[2][2];
    load_k(0, kb[0][0], kb[0][1]);
#pragma unroll
    for (int jt = 0; jt < SASA_NT; ++jt) {
        const int tile = wave + 4 * jt;
        sc[jt] = (sasa_f4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (jt + 1 < SASA_NT)
            load_k(jt + 1, kb[(jt + 1) & 1][0], kb[(jt + 1) & 1][1]);     // (clamped to a valid row past the end)
        __builtin_amdgcn_sched_barrier(0);
        if (tile < ntiles) {
            const rac_f4 k0 = kb[jt & 1][0], k1 = kb[jt & 1][1];
            sasa_f4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.x, qb4[0].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.y, qb4[0].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.z, qb4[0].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.w, qb4[0].w, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.x, qb4[1].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.y, qb4[1].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.z, qb4[1].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.w, qb4[1].w, acc, 0, 0, 0);
            // acc[r] = (q_query . k_key)/sqrt(d) for key = tile*16 + 4*lk + r, query = li; add the distance mask
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = tile * 16 + 4 * lk + r;
                if (kk < Q) {
                    const float dx = cqx - scen[2 * kk], dy = cqy - scen[2 * kk + 1];
                    const float v = acc[r] - sqrtf(dx * dx + dy * dy) * tau;
                    sc[jt][r] = v;
                    mloc = fmaxf(mloc, v);
                }
            }
        }
    }
    // per-query max: over lk inside the wave, then over the four waves
    mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    if (lk == 0)
        sred[wave * 16 + li] = mloc;
    __syncthreads();
    const float m = fmaxf(fmaxf(sred[li], sred[16 + li]), fmaxf(sred[32 + li], sred[48 + li]));
    float lsum = 0.f;
#pragma unroll
    for (int jt = 0; jt < SASA_NT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = (sc[jt][r] == -INFINITY) ? 0.f : expf(sc[jt][r] - m);
            sc[jt][r] = p;
            lsum += p;
        }
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    if (lk == 0)
        sred[64 + wave * 16 + li] = lsum;
    // O^T[ch][query] = sum_keys V[key][ch] * P^T[key][query]
    sasa_f4 oacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // (V rows one tile ahead as well)
    auto load_v = [&](int jt, float (&va)[2][4]) {
        const int tile = wave + 4 * jt;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = tile * 16 + 4 * lk + i;
            const int kc = key < Q ? key : Q - 1;
            const float *vp = a.qkv + (tok + kc) * ld + (2 * H + h) * SASA_D;
            va[0][i] = vp[li];
            va[1][i] = vp[16 + li];
        }
    };
    float vb[2][2][4];
    load_v(0, vb[0]);
#pragma unroll
    for (int jt = 0; jt < SASA_NT; ++jt) {
        const int tile = wave + 4 * jt;
        if (jt + 1 < SASA_NT)
            load_v(jt + 1, vb[(jt + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (tile < ntiles) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                oacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vb[jt & 1][0][i], sc[jt][i], oacc[0], 0, 0, 0);
                oacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vb[jt & 1][1][i], sc[jt][i], oacc[1], 0, 0, 0);
            }
        }
    }
    // combine the four waves: so[wave][ch][query]
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            so[(wave * 32 + 16 * c + 4 * lk + r) * 16 + li] = oacc[c][r];
    __syncthreads();
    {
        const int qq = tid >> 4, cp = (tid & 15) * 2;   // query within the tile, channel pair
        const int qrow = rb * SASA_ROWS + qq;
        if (qrow < Q) {
            const float l = (sred[64 + qq] + sred[64 + 16 + qq]) + (sred[64 + 32 + qq] + sred[64 + 48 + qq]);
            float o0 = 0.f, o1 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                o0 += so[(w * 32 + cp) * 16 + qq];
                o1 += so[(w * 32 + cp + 1) * 16 + qq];
            }
            *reinterpret_cast<float2 *>(a.out + (tok + qrow) * (size_t)(H * SASA_D) + h * SASA_D + cp) =
                make_float2(o0 / l, o1 / l);
        }
    }
}

extern "C" int rac_sasa_fwd(const float *qkv, const float *tau, const float *query_bbox, const float *box_table,
                            float *out, int ld_qkv, int ld_tau, int B, int Q, int heads, int dim, const float *pc_range,
                            void *stream)
{
    RAC_CHECK_ARG(dim == SASA_D, "rac_sasa_fwd: head dim %d (the kernel is built for %d)", dim, SASA_D);
    RAC_CHECK_ARG(B >= 0 && Q >= 0 && heads >= 1 && ld_tau >= heads && ld_qkv >= 3 * heads * dim && ld_qkv % 4 == 0, "rac_sasa_fwd: bad sizes B=%d Q=%d heads=%d", B, Q, heads);
    RAC_CHECK_ARG((size_t)Q * 2 * sizeof(float) <= 48 * 1024, "rac_sasa_fwd: Q=%d too large for the LDS centre table", Q);
    if (B == 0 || Q == 0)
        return 0;
    RAC_CHECK_ARG(qkv && tau && query_bbox && out && pc_range, "rac_sasa_fwd: null pointer");
    SasaArgs a;
    a.qkv = qkv; a.tau = tau; a.qbox = query_bbox; a.box = box_table; a.out = out;
    for (int i = 0; i < 6; ++i)
        a.pc[i] = pc_range[i];
    a.B = B; a.Q = Q; a.H = heads; a.ld_tau = ld_tau; a.ld_qkv = ld_qkv;
    a.row_blocks = (Q + SASA_ROWS - 1) / SASA_ROWS;
    const int nb = B * heads * a.row_blocks;
    const size_t cen = (size_t)2 * ((Q + 1) & ~1);
    if (Q <= 4 * SASA_NT * 16) {
        const size_t lds = (cen + 2 * 4 * 16 + 4 * 32 * 16) * sizeof(float);
        hipLaunchKernelGGL(sasa_mfma_kernel, dim3(nb), dim3(256), lds, (hipStream_t)stream, a);
    } else {
        const size_t lds = (cen + 2 * SASA_TILE * SASA_KS) * sizeof(float);
        hipLaunchKernelGGL(sasa_d32_kernel, dim3(nb), dim3(256), lds, (hipStream_t)stream, a);
    }
    return rac_launch_status("rac_sasa_fwd");
}
