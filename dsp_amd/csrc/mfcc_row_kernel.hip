// mfcc_row_kernel.hip -- gfx950 MFCC kernel, ROW-PER-FRAME form: one wave carries 4 frames,
// each on one 16-lane DPP row with 16 complex points per lane.
//
// Why a second form: the wave-per-frame kernel (mfcc_kernels.hip) is SIMD-issue bound
// (2 cycles per VALU op + ~2 cycles per dword moved to / from LDS), and a third
// of that goes to the three inter-stage exchanges a 4-points-per-lane FFT needs.  With 16
// points per lane the 256-point complex FFT is 16 x 16: two register-resident 16-point FFTs
// around ONE exchange through LDS (tools/emulate_row_fft.py), no permlane / DPP traffic.
//
//   load      16 x global_load_dwordx2 per lane: z[j+16k], rows = 4 consecutive frames
//   stage 1   fft16 over k (2 x radix-4 in registers), twiddle W256^(j q)
//   exchange  16 x ds_write_b64 / 8 x ds_read_b128, XOR swizzle, bank-conflict free
//   stage 2   fft16 over j -> lane q holds Z[q + 16 p]
//   untangle  natural-order Z image in LDS, lane (g,j) reads the pairs (k, 256-k), k = j+16m
//   tail      per frame as in the wave-per-frame kernel (sparse mel, log, DCT-II)
#include <hip/hip_runtime.h>

#include "mfcc_device.hpp"

namespace dsp {

namespace {

constexpr int ROW_TILE = 2048 + 256;                       // exchange tile / Z image / P of one row
constexpr int ROW_TILES_BYTES = 4 * ROW_TILE + 128;        // odd rows start 128 B later (bank spread)
constexpr int ROW_PART = ROW_TILES_BYTES;                  // 65 partial sums + zero slot
constexpr int ROW_LOGMEL = ROW_PART + 320;
constexpr int ROW_WAVE_BYTES = ROW_LOGMEL + 320;
static_assert(ROW_WAVE_BYTES % 16 == 0, "keep the carve 16-byte aligned");
// Block-shared constants behind the 4 wave regions, float4 [field][64 lanes]: what is used
// once per ITEM (4 frames) is re-read from LDS each time instead of occupying VGPRs
// (16 + 12 + 10 dwords per lane -> 10 ds_read_b128 per 4 frames).
// per-lane fields (x 64 lanes)
constexpr int CT_MEL = 0;      // 3 x float4: 12 mel weights
constexpr int CT_DCT = 3;      // up to 5 x float4: DCT weights
constexpr int CT_LANE_FIELDS = 8;
// per-j fields (x 16: the four rows of a wave read the same entries -> broadcast)
constexpr int CJ_TWP = 0;      // 4 x float4: W512^(j+16m), m = 0..7
constexpr int CJ_WIN = 4;      // 8 x float4: 32 window values            (DSP_ROW_CONST_LDS only)
constexpr int CJ_TW = 12;      // 8 x float4: W256^(j q), q = 1..15 (+pad)  (DSP_ROW_CONST_LDS only)
#ifndef DSP_ROW_CONST_LDS
#define DSP_ROW_CONST_LDS 0    // 1: window and W256 twiddles are also re-read from LDS per item (occupancy 3, measured slower)
#endif
constexpr int CJ_FIELDS = DSP_ROW_CONST_LDS ? 20 : 4;
constexpr int ROW_CTAB_LANE = 4 * ROW_WAVE_BYTES;
constexpr int ROW_CTAB_J = ROW_CTAB_LANE + CT_LANE_FIELDS * 64 * 16;
constexpr int ROW_BLOCK_BYTES = ROW_CTAB_J + CJ_FIELDS * 16 * 16;

// multiply by W16^1, W16^2, W16^3 (forward, exp(-2 pi i k / 16)) with literal constants
constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
__device__ __forceinline__ c32 mul_w16(c32 a, int k)
{
    switch (k & 15) {
    case 0: return a;
    case 1: return {a.x * C1 + a.y * S1, a.y * C1 - a.x * S1};
    case 2: return {(a.x + a.y) * R2, (a.y - a.x) * R2};
    case 3: return {a.x * S1 + a.y * C1, a.y * S1 - a.x * C1};
    case 4: return {a.y, -a.x};
    case 6: return {(a.y - a.x) * R2, -(a.x + a.y) * R2};
    case 9: return {-(a.x * C1 + a.y * S1), a.x * S1 - a.y * C1};   // -W16^1
    default: return a;   // not used
    }
}

// In-place 16-point DFT of v[0..15], 4 x 4, no second register set:
//   X[b + 4c] = sum_a W4^(ac) [ W16^(ab) sum_k1 v[a + 4 k1] W4^(k1 b) ]
// Pass 1 leaves t[a][b] in v[a + 4b]; pass 2 works on v[4b .. 4b+3] and leaves X[b + 4c] in
// v[c + 4b]: the result is in digit-swapped order, X[q] sits in v[DS(q)].
__host__ __device__ constexpr int DS(int q) { return (q >> 2) | ((q & 3) << 2); }

__device__ __forceinline__ void fft16_ds(c32 (&v)[16])
{
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        c32 x[4] = {v[a], v[a + 4], v[a + 8], v[a + 12]};
        radix4(x);
#pragma unroll
        for (int b = 0; b < 4; ++b) v[a + 4 * b] = mul_w16(x[b], a * b);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        c32 x[4] = {v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3]};
        radix4(x);
#pragma unroll
        for (int c = 0; c < 4; ++c) v[4 * b + c] = x[c];
    }
}

}  // namespace

template <int DCT_SPLIT, int DCT_LEN, int GATHER, bool FULL>
__global__ __launch_bounds__(256) void mfcc512_row_kernel(const Mfcc512Args args, const RowTables512 *__restrict__ R)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, j = lane & 15;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *wl = smem + wib * ROW_WAVE_BYTES;
    char *my_tile = wl + g * ROW_TILE + (g & 1) * 128;       // this lane's row
    float *part = reinterpret_cast<float *>(wl + ROW_PART);
    float *lmel = reinterpret_cast<float *>(wl + ROW_LOGMEL);

    const LaneTables512 *__restrict__ T = args.tables;

    // ---- per-lane constants ------------------------------------------------------------
#if !DSP_ROW_CONST_LDS
    float win[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) win[i] = R->win[i][lane];
    c32 tw[16];
#pragma unroll
    for (int q = 0; q < 15; ++q) tw[q] = {R->tw[2 * q][lane], R->tw[2 * q + 1][lane]};
#endif
    float4 *ctab = reinterpret_cast<float4 *>(smem + ROW_CTAB_LANE);
    float4 *ctabj = reinterpret_cast<float4 *>(smem + ROW_CTAB_J);
    if (threadIdx.x < 64) {       // one wave fills the block's tables
        const int l = threadIdx.x;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            ctab[(CT_MEL + i) * 64 + l] = make_float4(T->mel_w[4 * i][l], T->mel_w[4 * i + 1][l], T->mel_w[4 * i + 2][l], T->mel_w[4 * i + 3][l]);
#pragma unroll
        for (int i = 0; i < (DCT_LEN + 3) / 4; ++i)
            ctab[(CT_DCT + i) * 64 + l] = make_float4(T->dct_w[4 * i][l], T->dct_w[4 * i + 1][l], T->dct_w[4 * i + 2][l], T->dct_w[4 * i + 3][l]);
        if (l < 16) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                ctabj[(CJ_TWP + i) * 16 + l] = make_float4(R->twp[4 * i][l], R->twp[4 * i + 1][l], R->twp[4 * i + 2][l], R->twp[4 * i + 3][l]);
#if DSP_ROW_CONST_LDS
#pragma unroll
            for (int i = 0; i < 8; ++i)
                ctabj[(CJ_WIN + i) * 16 + l] = make_float4(R->win[4 * i][l], R->win[4 * i + 1][l], R->win[4 * i + 2][l], R->win[4 * i + 3][l]);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                ctabj[(CJ_TW + i) * 16 + l] = make_float4(R->tw[4 * i][l], R->tw[4 * i + 1][l], i < 7 ? R->tw[4 * i + 2][l] : 1.0f, i < 7 ? R->tw[4 * i + 3][l] : 0.0f);
#endif
        }
    }
    __syncthreads();
    const float4 *cj_lane = ctabj + j;
    const float4 *ct_lane = ctab + lane;
    const int mel_k0 = T->mel_k0[lane];
    int gat[GATHER];
#pragma unroll
    for (int q = 0; q < GATHER; ++q) gat[q] = T->mel_src[q][lane];
    const int n_mels = args.n_mels, n_mfcc = args.n_mfcc;
    constexpr int DCT_STRIDE = (DCT_LEN + 3) & ~3;
    const int dct_rd = (lane % DCT_SPLIT) * DCT_STRIDE;
    const int lmel_wr = (lane / DCT_LEN) * DCT_STRIDE + lane % DCT_LEN;
    const int dct_c = lane / DCT_SPLIT;
    const bool dct_store = (lane % DCT_SPLIT == 0) && dct_c < n_mfcc;

    // exchange addresses inside the row tile (bytes): element (slot q of lane j) at
    //   q*128 + ((j>>1) ^ (q>>1))*16 + (j&1)*8;  reader lane q' fetches pairs jj at q'*128 + (jj ^ (q'>>1))*16
    const int xw = (j >> 1) * 16 + (j & 1) * 8;             // ^ (q>>1)*16 + q*128 per slot
    const int xr = j * 128;                                  // + (jj ^ (j>>1))*16 per pair

    if (lane == 0) part[kZeroSlot] = 0.0f;
    if (lane < 16) lmel[64 + lane] = 0.0f;
    lmel[lane] = 0.0f;
    wave_lds_sync();

    const long wave = (long)blockIdx.x * 4 + wib;
    const long n_waves = (long)gridDim.x * 4;
    FrameCursor cur;
    cur.init(wave, n_waves, args.chunk, args.n_frames, args.frames_per_clip);
    if (!cur.valid()) return;

    const unsigned amin_u = __float_as_uint(args.amin);
    const float neg_top_db = -args.top_db;
    const int frame_len = args.frame_len;
    const long n_frames = args.n_frames;

    // this lane's frame is cur.f + g
    auto row_src = [&](const FrameCursor &c) -> const float * {
        if (args.frames_per_clip <= 0) return static_cast<const float *>(args.in) + (c.f + g) * (long)frame_len;
        long clip = c.clip;
        int t = c.t + g;
        while (t >= c.fpc) { t -= c.fpc; ++clip; }
        return static_cast<const float *>(args.in) + clip * args.clip_stride + (long)t * args.hop;
    };
    c32 v[16];
    auto load_item = [&](const FrameCursor &c) {
        const bool row_ok = c.f + g < n_frames;
        const float *src = row_src(c);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = 2 * (j + 16 * k);
            if (row_ok && (FULL || i + 1 < frame_len)) {
                const f2v x = args.frames_per_clip > 0 ? *reinterpret_cast<const f2v *>(src + i)      // clips re-read samples: cacheable
                                                       : __builtin_nontemporal_load(reinterpret_cast<const f2v *>(src + i));
                v[k] = {x.x, x.y};
            } else if (row_ok && i < frame_len) {
                v[k] = {src[i], 0.0f};
            } else {
                v[k] = {0.0f, 0.0f};
            }
        }
    };
    load_item(cur);

    while (true) {
        const long f0 = cur.f;
        // launder the table pointer once per item so the constant reads stay inside the loop
        const float4 *ct = ct_lane;
        const float4 *cj = cj_lane;
        asm volatile("" : "+v"(ct), "+v"(cj));
        // ---- window + stage 1 ---------------------------------------------------------
#if DSP_ROW_CONST_LDS
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 w4 = cj[(CJ_WIN + i) * 16];
            v[2 * i] = {v[2 * i].x * w4.x, v[2 * i].y * w4.y};
            v[2 * i + 1] = {v[2 * i + 1].x * w4.z, v[2 * i + 1].y * w4.w};
        }
        fft16_ds(v);                                          // A[q] in v[DS(q)]
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 t4 = cj[(CJ_TW + i) * 16];
            v[DS(2 * i + 1)] = cmul(v[DS(2 * i + 1)], c32{t4.x, t4.y});
            if (i < 7) v[DS(2 * i + 2)] = cmul(v[DS(2 * i + 2)], c32{t4.z, t4.w});
        }
#else
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = {v[k].x * win[2 * k], v[k].y * win[2 * k + 1]};
        fft16_ds(v);                                          // A[q] in v[DS(q)]
#pragma unroll
        for (int q = 1; q < 16; ++q) v[DS(q)] = cmul(v[DS(q)], tw[q - 1]);
#endif
        // ---- the one exchange: lane j slot q -> lane q slot j ---------------------------
#pragma unroll
        for (int q = 0; q < 16; ++q)
            *reinterpret_cast<float2 *>(my_tile + q * 128 + (xw ^ ((q >> 1) * 16))) = make_float2(v[DS(q)].x, v[DS(q)].y);
        wave_lds_sync();
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const float4 x = *reinterpret_cast<const float4 *>(my_tile + xr + ((jj ^ (j >> 1)) * 16));
            v[2 * jj] = {x.x, x.y};
            v[2 * jj + 1] = {x.z, x.w};
        }
        wave_lds_sync();
        fft16_ds(v);                                          // Z[j + 16 p] / 2 in v[DS(p)]
        // ---- natural-order Z image (Z[256] = Z[0]) ---------------------------------------
#pragma unroll
        for (int p = 0; p < 16; ++p)
            *reinterpret_cast<float2 *>(my_tile + (j + 16 * p) * 8) = make_float2(v[DS(p)].x, v[DS(p)].y);
        if (j == 0) *reinterpret_cast<float2 *>(my_tile + 256 * 8) = make_float2(v[0].x, v[0].y);
        wave_lds_sync();

        cur.next(4);
        const bool more = cur.valid();
        c32 u[8], w[8];
        float z128x = 0.f, z128y = 0.f;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const float2 a = *reinterpret_cast<const float2 *>(my_tile + (j + 16 * m) * 8);
            const float2 b = *reinterpret_cast<const float2 *>(my_tile + (256 - j - 16 * m) * 8);
            u[m] = {a.x, a.y};
            w[m] = {b.x, b.y};
        }
        {
            const float2 c = *reinterpret_cast<const float2 *>(my_tile + 128 * 8);
            z128x = c.x; z128y = c.y;
        }
        wave_lds_sync();
        c32 twp[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 t4 = cj[(CJ_TWP + i) * 16];
            twp[2 * i] = {t4.x, t4.y};
            twp[2 * i + 1] = {t4.z, t4.w};
        }

        // ---- untangle + power; P goes back into the row tile as floats ---------------------
        float *prow = reinterpret_cast<float *>(my_tile);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const c32 E = {u[m].x + w[m].x, u[m].y - w[m].y};
            const c32 O = {u[m].x - w[m].x, u[m].y + w[m].y};
            const c32 Tw = cmul(O, twp[m]);
            const float xr_ = E.x + Tw.y, xi = E.y - Tw.x;
            const float mr = E.x - Tw.y, mi = E.y + Tw.x;
            prow[j + 16 * m] = xr_ * xr_ + xi * xi;
            prow[256 - j - 16 * m] = mr * mr + mi * mi;
        }
        if (j == 0) prow[128] = 4.0f * (z128x * z128x + z128y * z128y);
        wave_lds_sync();
        if (more) load_item(cur);        // the 32 data registers refill under the tail
        float melw[kMelChunk], dctw[(DCT_LEN + 3) & ~3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float4 t4 = ct[(CT_MEL + i) * 64];
            melw[4 * i] = t4.x; melw[4 * i + 1] = t4.y; melw[4 * i + 2] = t4.z; melw[4 * i + 3] = t4.w;
        }
#pragma unroll
        for (int i = 0; i < (DCT_LEN + 3) / 4; ++i) {
            const float4 t4 = ct[(CT_DCT + i) * 64];
            dctw[4 * i] = t4.x; dctw[4 * i + 1] = t4.y; dctw[4 * i + 2] = t4.z; dctw[4 * i + 3] = t4.w;
        }

        // ---- tail, one frame (row) at a time, all 64 lanes -----------------------------------
#pragma unroll 1
        for (int r = 0; r < 4; ++r) {
            if (f0 + r >= n_frames) break;                     // wave-uniform
            const float *pbuf = reinterpret_cast<const float *>(wl + r * ROW_TILE + (r & 1) * 128);
            const float *rd = pbuf + mel_k0;
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < kMelChunk; ++i) acc = fmaf(melw[i], rd[i], acc);
            part[lane] = acc;
            wave_lds_sync();
            float e = part[gat[0]];
#pragma unroll
            for (int q = 1; q < GATHER; ++q) e += part[gat[q]];
            if (lane >= n_mels) e = 0.0f;
            const float ref = __uint_as_float(max(__float_as_uint(wave_max_nonneg(e)), amin_u));
            const float ec = __uint_as_float(max(__float_as_uint(e), amin_u));
            const float k10 = 3.01029995663981195f;
            float db = k10 * __builtin_amdgcn_logf(ec * __builtin_amdgcn_rcpf(ref));
            db = __builtin_amdgcn_fmed3f(db, neg_top_db, 0.0f);
            if (lane < n_mels) lmel[lmel_wr] = db;
            wave_lds_sync();
            const float *dr = lmel + dct_rd;
            float c = 0.0f;
#pragma unroll
            for (int i = 0; i + 4 <= DCT_LEN; i += 4) {
                const float4 x = *reinterpret_cast<const float4 *>(dr + i);
                c = fmaf(dctw[i], x.x, c);
                c = fmaf(dctw[i + 1], x.y, c);
                c = fmaf(dctw[i + 2], x.z, c);
                c = fmaf(dctw[i + 3], x.w, c);
            }
            if (DCT_LEN % 4) {
                const float2 x = *reinterpret_cast<const float2 *>(dr + (DCT_LEN & ~3));
                c = fmaf(dctw[DCT_LEN & ~3], x.x, c);
                c = fmaf(dctw[(DCT_LEN & ~3) + 1], x.y, c);
            }
            c += dpp<DPP_QUAD_1032>(c);
            if (DCT_SPLIT == 4) c += dpp<DPP_QUAD_2301>(c);
            if (dct_store) args.out[(f0 + r) * n_mfcc + dct_c] = c;
            wave_lds_sync();
        }
        if (!more) return;
    }
}

hipError_t launch_mfcc512_row(const Mfcc512Args &args, const RowTables512 *row_tables, int dct_split, int dct_len,
                              int gather, int blocks, hipStream_t stream)
{
    const bool full = args.frame_len == 512;
    const size_t lds = (size_t)ROW_BLOCK_BYTES;
#define DSP_LAUNCH(S, L, G)                                                                                          \
    if (dct_split == S && dct_len == L && gather == G) {                                                             \
        if (full) hipLaunchKernelGGL((mfcc512_row_kernel<S, L, G, true>), dim3(blocks), dim3(256), lds, stream, args, row_tables);   \
        else hipLaunchKernelGGL((mfcc512_row_kernel<S, L, G, false>), dim3(blocks), dim3(256), lds, stream, args, row_tables);  \
        return hipGetLastError();                                                                                    \
    }
    DSP_LAUNCH(4, 10, 3)
    DSP_LAUNCH(4, 10, 6)
    DSP_LAUNCH(4, 16, 3)
    DSP_LAUNCH(4, 16, 6)
    DSP_LAUNCH(2, 20, 3)
    DSP_LAUNCH(2, 20, 6)
#undef DSP_LAUNCH
    return hipErrorInvalidConfiguration;
}

int mfcc512_row_blocks_per_cu(int dct_split, int dct_len, int gather, bool full)
{
    int n = 0;
    const size_t lds = (size_t)ROW_BLOCK_BYTES;
#define DSP_OCC(S, L, G)                                                                                   \
    if (dct_split == S && dct_len == L && gather == G) {                                                   \
        hipError_t e = full ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, mfcc512_row_kernel<S, L, G, true>, 256, lds)   \
                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, mfcc512_row_kernel<S, L, G, false>, 256, lds); \
        return e == hipSuccess && n > 0 ? n : 3;                                                           \
    }
    DSP_OCC(4, 10, 3)
    DSP_OCC(4, 10, 6)
    DSP_OCC(4, 16, 3)
    DSP_OCC(4, 16, 6)
    DSP_OCC(2, 20, 3)
    DSP_OCC(2, 20, 6)
#undef DSP_OCC
    return 3;
}

}  // namespace dsp
