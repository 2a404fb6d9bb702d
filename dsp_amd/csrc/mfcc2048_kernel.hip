// mfcc2048_kernel.hip -- the MFCC chain at n_fft = 2048: the framing of the file north_star names, cepstrum/scrubjay_infer.c
// (:10-14: WIN_SIZE 2048, HOP_SIZE 1024, N_FILTERS 40, N_MFCC 20; pooling :36-66; SVM :105-141).  Same chain as the other
// kernels (reference 2fa/audio/word/c/mfcc.c:142-221 with the constants as parameters), one 64-lane wavefront per frame:
//
//   load      16 x global_load_dwordx2 per lane: z[l + 64 a] = x[2n] + i x[2n+1] (zero padded past frame_length), x window / 2
//   FFT       2048-point real FFT as a 1024-point complex Stockham autosort in three passes, 16 x 16 x 4 (a radix-16 butterfly
//             per lane, then four radix-4 ones), through an 8.5 KB padded per-wave LDS image, twiddles from a block-shared LDS table
//   untangle  bins k = l + 64 t (t < 8) with 1024 - k; power spectrum P[0..1024] to LDS
//   mel       every lane dots three 16-bin segments of the filters' runs (weights in registers), the filter's lane adds its
//             segments in ascending order; banks that do not fit 192 segments fall back to one lane per filter (CSR walk)
//   log       per-frame reference = max, amin, top_db (mfcc.c:169-206)
//   DCT-II    two lanes per coefficient (<= 32 coefficients), halves of the log-mel vector each, rows from a block-shared LDS copy
//   POOL      instead of storing the coefficients: per-clip mean | std in float64, frame order (scrubjay_infer.c:36-66), then
//             Scaler -> RBF-SVM -> libsvm's label / probability (svm_kernels.hpp), one wavefront walks one clip
//
// This shape exists so that the fused clip -> label path can run scrubjay_infer.c's own parameterisation.  Round 2 tuned the
// obvious: the mel product in 16-bin segments on all 64 lanes with the weights in registers (a lane-per-filter walk of up to
// 135 weights from global memory took 30 % of the kernel), the DCT rows from LDS, three 16 x 16 x 4 passes instead of five
// radix-4 ones: 10.4 -> 5.75 ms per 125 000 one-second clips (fused), 1.14 -> 0.82 ms per 250 000 plain frames.  Rising wave
// priorities (mfcc_kernels.hip) were tried here too and lose 1.5 % at two waves per SIMD.
#include <hip/hip_runtime.h>

#include "mfcc_device.hpp"
#include "svm_kernels.hpp"
#include "tables.hpp"

namespace dsp {

namespace {

constexpr int Q_ZBUF = 0;                          // 1025 x float2 image (Z[1024] = Z[0]), one float2 of padding every 16; later P[0..1024]
constexpr int Q_LMEL = 1092 * 8;                   // 128 log-mel values
__device__ __forceinline__ constexpr int ZI(int i) { return i + (i >> 4); }           // padded index into the image

// forward 16-point DFT in place as 4 x 4 (input t = 4 t1 + t2): position p = 4 q1 + q2 ends up holding X[q1 + 4 q2] = X[r16_out(p)]
__device__ __forceinline__ constexpr int r16_out(int p) { return (p >> 2) + 4 * (p & 3); }
__device__ __forceinline__ void radix16(c32 (&v)[16])
{
#pragma unroll
    for (int t2 = 0; t2 < 4; ++t2) {                // DFT over t1 for each t2 -> A[t2][q1] at v[4 q1 + t2]
        c32 u[4] = {v[t2], v[4 + t2], v[8 + t2], v[12 + t2]};
        radix4(u);
#pragma unroll
        for (int q1 = 0; q1 < 4; ++q1) v[4 * q1 + t2] = u[q1];
    }
    // W16^(t2 q1) = exp(-2 pi i t2 q1 / 16)
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
    constexpr c32 W1 = {C1, -S1}, W2 = {R2, -R2}, W3 = {S1, -C1}, W6 = {-R2, -R2}, W9 = {-C1, S1};
    v[4 * 1 + 1] = cmul(v[4 * 1 + 1], W1); v[4 * 1 + 2] = cmul(v[4 * 1 + 2], W2); v[4 * 1 + 3] = cmul(v[4 * 1 + 3], W3);
    v[4 * 2 + 1] = cmul(v[4 * 2 + 1], W2); v[4 * 2 + 2] = cmul_mi(v[4 * 2 + 2]);   v[4 * 2 + 3] = cmul(v[4 * 2 + 3], W6);
    v[4 * 3 + 1] = cmul(v[4 * 3 + 1], W3); v[4 * 3 + 2] = cmul(v[4 * 3 + 2], W6); v[4 * 3 + 3] = cmul(v[4 * 3 + 3], W9);
#pragma unroll
    for (int q1 = 0; q1 < 4; ++q1) {                // DFT over t2 for each q1 -> X[q1 + 4 q2] at v[4 q1 + q2]
        c32 u[4] = {v[4 * q1], v[4 * q1 + 1], v[4 * q1 + 2], v[4 * q1 + 3]};
        radix4(u);
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) v[4 * q1 + q2] = u[q2];
    }
}
constexpr int Q_FEAT = Q_LMEL + 128 * 4;           // POOL: 64 standardised features
constexpr int Q_PART = Q_FEAT + 64 * 4;            // partial sums of the mel segments (192) + the slot that reads 0
constexpr int Q_WAVE_BYTES = Q_PART + 196 * 4;
static_assert(Q_WAVE_BYTES % 16 == 0, "keep the carve 16-byte aligned");
constexpr int Q_W1024 = 4 * Q_WAVE_BYTES;          // block-shared: W1024^i, i < 1024
constexpr int Q_W2048 = Q_W1024 + 1024 * 8;        // block-shared: W2048^k, k < 512
constexpr int Q_TW1 = Q_W2048 + 512 * 8;           // block-shared: pass 1's twiddles W_256^(t k) at [t - 1][k], t = 1 .. 15, k < 16
constexpr int Q_SEGW = Q_TW1 + 15 * 16 * 8;        // block-shared, PRE kernels only: the mel segments' weights seg_w[c][i][lane] (12 KB; out of 48 VGPRs per lane)
constexpr int Q_SEGW_BYTES = k2048SegSlots * k2048SegTaps * 64 * 4;
// PRE (= the aubio-semantics kernels, 40 filters: 81.5 KB per block, two blocks per CU still fit; the 128-filter librosa plans would not):
// the filterbank weights live in LDS and the registers they held carry the next frame's samples, requested one frame ahead
__device__ __host__ constexpr int q_win(bool pre) { return Q_SEGW + (pre ? Q_SEGW_BYTES : 0); }      // block-shared: window pairs (win[2a][l], win[2a+1][l]) at [a][l]
__device__ __host__ constexpr int q_dct(bool pre) { return q_win(pre) + 16 * 64 * 8; }                // block-shared: dct_t[i][lane], i < ceil(n_mels / 2)
// + 256 B per DCT row (+ POOL: 2 x 64 floats, the Scaler's offset | scale), added by the launcher

}  // namespace

// AUB: the aubio-semantics front end of cepstrum/scrubjay_infer.c:21-53 (dsp_mfcc_scrubjay_infer_config; restated from aubio
// 0.4's published algorithm, parity unpinned): any of
//   args.spectrum = 1         the MAGNITUDE |X[k]| goes into the filterbank (aubio_fft_get_norm + aubio_filterbank_do, power 1)
//   args.log_mode = 2         plain log10 of each filter output, inputs below 2e-42 count as 2e-42 (fvec_log10 / SAFE_LOG10)
//   args.stream_framing       aubio_source_do + aubio_pvoc_do: frame t of a clip = samples [(t + 1) hop - frame_len, (t + 1) hop),
//                             zeros before the clip (the vocoder's empty history) and past its end (the short last read)
// The filterbank (DSP_MELNORM_AUBIO_SLANEY) is a table like any other.  A separate instantiation: the reference-semantics
// kernels keep their code and registers.
// IN: 0 float samples; 1 int16 mono, 2 interleaved int16 stereo channel 0, 3 stereo (L + R) / 65536 -- converted
// in the load (exact: the float path's values), instantiated for the scrubjay_infer.c front end (AUB), whose callers decode int16 files.
template <int IN>
__device__ __forceinline__ void load_pair_2048(const void *base, long i, bool cached, float &x0, float &x1)
{
    if constexpr (IN == 0) {
        const f2v x = cached ? *reinterpret_cast<const f2v *>(static_cast<const float *>(base) + i)
                             : __builtin_nontemporal_load(reinterpret_cast<const f2v *>(static_cast<const float *>(base) + i));
        x0 = x.x; x1 = x.y;
    } else if constexpr (IN == 1) {
        const unsigned q = *reinterpret_cast<const unsigned *>(static_cast<const short *>(base) + i);
        x0 = (float)(int)(short)(q & 0xffffu) * (1.0f / 32768.0f); x1 = (float)((int)q >> 16) * (1.0f / 32768.0f);
    } else {
        const uint2 q = *reinterpret_cast<const uint2 *>(static_cast<const short *>(base) + 2 * i);
        if constexpr (IN == 2) { x0 = (float)(int)(short)(q.x & 0xffffu) * (1.0f / 32768.0f); x1 = (float)(int)(short)(q.y & 0xffffu) * (1.0f / 32768.0f); }
        else { x0 = (float)((int)(short)(q.x & 0xffffu) + ((int)q.x >> 16)) * (1.0f / 65536.0f); x1 = (float)((int)(short)(q.y & 0xffffu) + ((int)q.y >> 16)) * (1.0f / 65536.0f); }
    }
}
template <int IN>
__device__ __forceinline__ float load_one_2048(const void *base, long i)
{
    if constexpr (IN == 0) return static_cast<const float *>(base)[i];
    else if constexpr (IN == 1) return (float)static_cast<const short *>(base)[i] * (1.0f / 32768.0f);
    else if constexpr (IN == 2) return (float)static_cast<const short *>(base)[2 * i] * (1.0f / 32768.0f);
    else return (float)((int)static_cast<const short *>(base)[2 * i] + (int)static_cast<const short *>(base)[2 * i + 1]) * (1.0f / 65536.0f);
}

template <bool CLIPS, bool POOL, bool AUB = false, int IN = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void mfcc2048_kernel(const Mfcc512Args args, const GenTables2048 *__restrict__ G)
{
    static_assert(!POOL || CLIPS, "pooling is per clip");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *wl = smem + wib * Q_WAVE_BYTES;
    float2 *zbuf = reinterpret_cast<float2 *>(wl + Q_ZBUF);
    float *pbuf = reinterpret_cast<float *>(wl + Q_ZBUF);
    float *lmel = reinterpret_cast<float *>(wl + Q_LMEL);
    float *feat = reinterpret_cast<float *>(wl + Q_FEAT);
    float2 *w1024 = reinterpret_cast<float2 *>(smem + Q_W1024);
    float2 *w2048 = reinterpret_cast<float2 *>(smem + Q_W2048);
    float *part = reinterpret_cast<float *>(wl + Q_PART);
    constexpr bool PRE = AUB;
    float *dct_t = reinterpret_cast<float *>(smem + q_dct(PRE));
    float2 *win2 = reinterpret_cast<float2 *>(smem + q_win(PRE));
    float2 *tw1 = reinterpret_cast<float2 *>(smem + Q_TW1);
    float *segw = reinterpret_cast<float *>(smem + Q_SEGW);
    const int n_mels = args.n_mels, n_mfcc = args.n_mfcc;
    const int half = (n_mels + 1) / 2;                   // log-mels per DCT lane
    for (int i = threadIdx.x; i < 1024; i += 256) w1024[i] = make_float2(G->w1024[0][i], G->w1024[1][i]);
    for (int i = threadIdx.x; i < 512; i += 256) w2048[i] = make_float2(G->w2048[0][i], G->w2048[1][i]);
    // pass 1 reads W_1024^(4 t k), k = lane % 16: out of the full table that is a stride of 8 t dwords -- 16 addresses on 8 / 4 / 2 / 1
    // banks for t odd / 2 mod 4 / 4 mod 8 / 8 (tools/lds_banks_2048.py: 128 LDS cycles per frame where 30 would do); its own table
    // [t - 1][k] puts the 16 values of a t side by side
    if (threadIdx.x < 240) { const int i = 4 * (threadIdx.x / 16 + 1) * (threadIdx.x % 16); tw1[threadIdx.x] = make_float2(G->w1024[0][i], G->w1024[1][i]); }
    for (int i = threadIdx.x; i < half * 64; i += 256) dct_t[i] = (&G->dct_t[0][0])[i];
    // the window through LDS pays in the fused clip kernel (+4 %) and costs the plain one dearly (0.82 -> 1.34 ms per 250 000
    // frames, measured): it stays a global (L1) read there
    constexpr bool WIN_LDS = POOL;
    if (WIN_LDS)
        for (int i = threadIdx.x; i < 16 * 64; i += 256) win2[i] = make_float2(G->win[2 * (i >> 6)][i & 63], G->win[2 * (i >> 6) + 1][i & 63]);
    // POOL: the Scaler's offset | scale in a block-shared LDS copy behind the DCT rows (the clip's tail read them with dependent
    // global loads, each behind an s_waitcnt that also drains the frame's loads; mfcc_kernels.hip, svm_small)
    float *scaler = dct_t + half * 64;
    (void)scaler;
    if (POOL)
        for (int i = threadIdx.x; i < 2 * n_mfcc; i += 256) { scaler[i] = args.pool.svm.offset[i]; scaler[64 + i] = args.pool.svm.scale[i]; }
    if (lane == 0) part[k2048SegZero] = 0.0f;
    __syncthreads();

    // mel segments: this lane's three 16-bin windows and their weights stay in registers; filters lane and lane + 64 add
    // their partial sums [s0, s0 + cnt)
    const bool seg_ok = G->seg_ok != 0;
    // The filterbank's weights used to ride in 48 registers per lane across the frame loop; they are read from a block-shared LDS copy
    // now (conflict-free: consecutive lanes, consecutive words), and the registers carry the NEXT frame's samples instead (request_frame)
    int seg_k0[k2048SegSlots];
    float seg_w[PRE ? 1 : k2048SegSlots][PRE ? 1 : k2048SegTaps];      // !PRE: in registers, as before
    int mel_s0[2], mel_cnt[2];
#pragma unroll
    for (int c = 0; c < k2048SegSlots; ++c) {
        seg_k0[c] = G->seg_k0[c][lane];
#pragma unroll
        for (int i = 0; i < k2048SegTaps; ++i) {
            if constexpr (PRE) segw[(c * k2048SegTaps + i) * 64 + lane] = G->seg_w[c][i][lane];      // (every wave writes the same values; a wave reads what its own lanes wrote)
            else seg_w[c][i] = G->seg_w[c][i][lane];
        }
    }
    (void)seg_w; (void)segw;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = lane + 64 * i;
        mel_s0[i] = m < n_mels ? G->mel_s0[m] : 0;
        mel_cnt[i] = m < n_mels ? G->mel_cnt[m] : 0;
    }

    const long wave = (long)blockIdx.x * 4 + wib;
    const long n_waves = (long)gridDim.x * 4;
    const unsigned amin_u = __float_as_uint(args.amin);
    const float neg_top_db = -args.top_db;
    const int frame_len = args.frame_len;
    (void)feat;

    // (the fused clip kernel walks whole clips, uniform or ragged: ClipCursor)
    std::conditional_t<POOL, ClipCursor, WaveCursor<CLIPS>> cur;
    if constexpr (POOL) cur.init(wave, n_waves, args.n_clips, args.frames_per_clip, args.hop, args.clip_stride, args.spans, args.samples_per_clip);
    else cur.init(wave, n_waves, args.chunk, args.n_frames, args.frames_per_clip, CLIPS ? args.hop : frame_len, args.clip_stride);
    if (!cur.valid()) return;

    // POOL: lane 2 c keeps the running sums of coefficient c for the clip this wave is walking
    double pool_s = 0.0, pool_q = 0.0;
    int pool_t = 0;
    (void)pool_s; (void)pool_q; (void)pool_t;

    // The frame's samples are requested ONE FRAME AHEAD, from the middle of the frame before it (behind the power spectrum, where the
    // transform's 32 registers are free again): at the loop's top they stood, with the whole HBM latency, in front of every frame's
    // first instruction -- a wave spent 35 % of its cycles in s_waitcnt (SQ_WAIT_ANY), most of it there, and two waves per SIMD do
    // not cover that for each other.  raw[a] = samples 2 (lane + 64 a), + 1 of the frame, zero outside [lo_i, hi_i).
    c32 raw[16];
    auto request_frame = [&]() {         // the frame `cur` stands at (cur.valid())
        long src = cur.off;                                              // first sample of the frame (samples per channel from args.in)
        // samples [lo_i, hi_i) of the frame exist; the rest reads as zero.  Complete frames: [0, frame_len).
        int lo_i = 0, hi_i = frame_len;
        if (AUB && CLIPS && args.stream_framing) {
            const int start = (cur.t + 1) * args.hop - frame_len;        // first sample of the frame inside its clip (even; < 0: history)
            src += args.hop - frame_len;                                 // cur.off = clip_off + t * hop
            lo_i = start < 0 ? -start : 0;
            int n_clip = args.samples_per_clip;                          // ragged batches: the clip's own length
            if constexpr (POOL) n_clip = cur.n_samples;
            hi_i = min(frame_len, n_clip - start);
        }
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            const int i = 2 * (lane + 64 * a);
            float x0 = 0.0f, x1 = 0.0f;
            if (i + 1 < hi_i && (!AUB || i >= lo_i)) {                   // lo_i is even: a pair never straddles it
                load_pair_2048<IN>(args.in, src + i, CLIPS, x0, x1);     // clips re-read samples: cacheable
            } else if (i < hi_i && (!AUB || i >= lo_i)) {
                x0 = load_one_2048<IN>(args.in, src + i);
            }
            raw[a] = {x0, x1};
        }
    };
    if constexpr (PRE) request_frame();

    while (cur.valid()) {
        const long f = cur.f, clip_f = cur.clip;
        const bool last_of_chunk = cur.left == 0 || cur.remaining == 1;
        (void)clip_f; (void)last_of_chunk;
        if constexpr (!PRE) request_frame();                             // (registers are tight there: the samples are loaded where they are used)
        cur.next();

        // ---- window: v[a] = z[lane + 64 a] --------------------------------------------------------------------
        c32 v[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            if (WIN_LDS) {
                const float2 w = win2[64 * a + lane];
                v[a] = {raw[a].x * w.x, raw[a].y * w.y};
            } else {
                v[a] = {raw[a].x * G->win[2 * a][lane], raw[a].y * G->win[2 * a + 1][lane]};
            }
        }

        // ---- 1024-point complex FFT: Stockham autosort as 16 x 16 x 4, three passes through the wave's LDS image -------------
        // pass with Ns points done, butterfly j: k = j % Ns, inputs x[j + (1024 / R) t] W_{R Ns}^(t k), outputs y[(j - k) R + k + q Ns].
        // The image is padded by one float2 every 16 (ZI): the stride-16 stores of the radix-16 passes would otherwise all land in
        // one bank pair.
        {
            // pass 0: R = 16, Ns = 1: butterfly j = lane takes the 16 points this lane loaded, no twiddles
            radix16(v);
#pragma unroll
            for (int p = 0; p < 16; ++p) zbuf[ZI(16 * lane + r16_out(p))] = make_float2(v[p].x, v[p].y);
            wave_lds_sync();
            // pass 1: R = 16, Ns = 16: k = lane % 16, twiddles W_256^(t k) = W_1024^(4 t k)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float2 x = zbuf[ZI(lane + 64 * t)];
                v[t] = {x.x, x.y};
            }
            wave_lds_sync();
            const int k = lane & 15;
#pragma unroll
            for (int t = 1; t < 16; ++t) {
                const float2 w = tw1[16 * (t - 1) + k];
                v[t] = cmul(v[t], c32{w.x, w.y});
            }
            radix16(v);
            const int base = 16 * (lane - k) + k;
#pragma unroll
            for (int p = 0; p < 16; ++p) zbuf[ZI(base + 16 * r16_out(p))] = make_float2(v[p].x, v[p].y);
            wave_lds_sync();
            // pass 2: R = 4, Ns = 256: four butterflies per lane, j = lane + 64 m = k, outputs y[j + 256 q]: natural order
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float2 x = zbuf[ZI(lane + 64 * m + 256 * t)];
                    v[m + 4 * t] = {x.x, x.y};
                }
            wave_lds_sync();
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int j = lane + 64 * m;
                c32 u[4] = {v[m], v[m + 4], v[m + 8], v[m + 12]};
#pragma unroll
                for (int t = 1; t < 4; ++t) {
                    const float2 w = w1024[(t * j) & 1023];
                    u[t] = cmul(u[t], c32{w.x, w.y});
                }
                radix4(u);
#pragma unroll
                for (int q = 0; q < 4; ++q) zbuf[ZI(j + 256 * q)] = make_float2(u[q].x, u[q].y);
            }
            wave_lds_sync();
        }
        if (lane == 0) zbuf[ZI(1024)] = zbuf[ZI(0)];         // natural order now; Z[1024] = Z[0] for the pairing below
        wave_lds_sync();

        // ---- untangle: bins k = l + 64 t (t < 8) with 1024 - k; bin 512 alone -------------------------------------------------
        float P[16];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int k = lane + 64 * t;
            const float2 a = zbuf[ZI(k)], b = zbuf[ZI(1024 - k)], w = w2048[k];
            const c32 E = {a.x + b.x, a.y - b.y};
            const c32 O = {a.x - b.x, a.y + b.y};
            const c32 Tw = cmul(O, c32{w.x, w.y});
            const float xr = E.x + Tw.y, xi = E.y - Tw.x;
            const float mr = E.x - Tw.y, mi = E.y + Tw.x;
            P[2 * t] = xr * xr + xi * xi;
            P[2 * t + 1] = mr * mr + mi * mi;
        }
        const bool magnitude = AUB && args.spectrum != 0;
        if (magnitude) {
#pragma unroll
            for (int t = 0; t < 16; ++t) P[t] = __builtin_amdgcn_sqrtf(P[t]);       // |X[k]|: aubio_fft_get_norm
        }
        const float2 zm = zbuf[ZI(512)];
        wave_lds_sync();
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            pbuf[lane + 64 * t] = P[2 * t];
            pbuf[1024 - lane - 64 * t] = P[2 * t + 1];
        }
        if (lane == 0) {
            const float p512 = 4.0f * (zm.x * zm.x + zm.y * zm.y);
            pbuf[512] = magnitude ? __builtin_amdgcn_sqrtf(p512) : p512;
        }
        wave_lds_sync();
        if constexpr (PRE) { if (cur.valid()) request_frame(); }         // the next frame's samples: in flight during mel, log, DCT and the clip's tail

        // ---- mel: lane m (and m + 64) walks filter m's run of weights in ascending bins (mfcc.c:158-164) --------------------
        float e[2] = {0.0f, 0.0f};
        float emax = 0.0f;
        if (seg_ok) {
            // every lane dots its three windows (ascending bins), the filter's lane adds the pieces in ascending order
#pragma unroll
            for (int c = 0; c < k2048SegSlots; ++c) {
                const float *rd = pbuf + seg_k0[c];
                float acc = 0.0f;
#pragma unroll
                for (int i = 0; i < k2048SegTaps; ++i) {
                    if constexpr (PRE) acc = fmaf(segw[(c * k2048SegTaps + i) * 64 + lane], rd[i], acc);
                    else acc = fmaf(seg_w[c][i], rd[i], acc);
                }
                part[c * 64 + lane] = acc;
            }
            wave_lds_sync();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (i == 1 && n_mels <= 64) break;
                float acc = 0.0f;
#pragma unroll
                for (int g = 0; g < k2048MaxGather; ++g) acc += part[g < mel_cnt[i] ? mel_s0[i] + g : k2048SegZero];
                e[i] = acc;
                emax = fmaxf(emax, acc);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = lane + 64 * i;
                float acc = 0.0f;
                if (m < n_mels) {
                    const int lo = G->mel_lo[m], len = G->mel_len[m];
                    const float *w = G->mel_w + G->mel_off[m];
                    for (int k = 0; k < len; ++k) acc = fmaf(w[k], pbuf[lo + k], acc);
                }
                e[i] = acc;
                emax = fmaxf(emax, acc);
            }
        }
        const bool aub_log = AUB && args.log_mode == 2;
        if (aub_log) {
            // aubio_mfcc_do: fvec_log10 = log10(x < 2e-42 ? 2e-42 : x), no reference, no floor relative to the frame.  v_log_f32
            // does not take denormal inputs: values below 2^-100 are scaled into range first.
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float v = e[i];
                const bool tiny = v < 0x1p-100f;
                const float l2 = __builtin_amdgcn_logf(tiny ? v * 0x1p+64f : v) - (tiny ? 64.0f : 0.0f);
                const float lg = v < 2e-42f ? -41.69897f : 0.30102999566398120f * l2;       // log10(2e-42) = -41.69897
                lmel[lane + 64 * i] = (lane + 64 * i < n_mels) ? lg : 0.0f;
            }
        }
        // ---- 10 log10 with per-frame reference (mfcc.c:169-206), one log of the ratio --------------------------------------
        const float ref = aub_log ? 0.0f : __uint_as_float(max(__float_as_uint(wave_max_nonneg(emax)), amin_u));
        if (aub_log) {
        } else if (!POOL && args.log_mode != 0) {
            // librosa power_to_db(ref = 1.0, top_db below the CLIP's maximum), keyword_classifier.py:42-55 / librosa's defaults as
            // cepstrum/train.py:45-52 uses them; same two passes as the 512-point kernel (mfcc_kernels.hip)
            const float k10 = 3.01029995663981195f;
            const float top_db_val = k10 * __builtin_amdgcn_logf(ref);          // this frame's maximum in dB
            if (args.frame_max != nullptr) {                                    // pass 1: only the frame maximum
                if (lane == 0) args.frame_max[f] = top_db_val;
                wave_lds_sync();
                continue;
            }
            const float floor_db = args.clip_floor ? args.clip_floor[clip_f] : top_db_val + neg_top_db;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float ec = __uint_as_float(max(__float_as_uint(e[i]), amin_u));
                const float db = fmaxf(k10 * __builtin_amdgcn_logf(ec), floor_db);
                lmel[lane + 64 * i] = (lane + 64 * i < n_mels) ? db : 0.0f;
            }
        } else {
            const float inv = __builtin_amdgcn_rcpf(ref);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float ec = __uint_as_float(max(__float_as_uint(e[i]), amin_u));
                float db = 3.01029995663981195f * __builtin_amdgcn_logf(ec * inv);
                db = __builtin_amdgcn_fmed3f(db, neg_top_db, 0.0f);
                lmel[lane + 64 * i] = (lane + 64 * i < n_mels) ? db : 0.0f;          // 128 slots: the DCT may read up to 2 half <= n_mels + 1
            }
        }
        wave_lds_sync();

        // ---- DCT-II: lane 2 c + h dots log-mels [h half, h half + half) with its column of the LDS copy of dct_t -------------
        float coef;
        {
            const int c = lane >> 1, h = lane & 1;
            const float *lm = lmel + h * half;
            float acc = 0.0f;
#pragma unroll 4
            for (int m = 0; m < half; ++m) acc = fmaf(dct_t[m * 64 + lane], lm[m], acc);     // rows past n_mels / n_mfcc hold 0
            acc += dpp<DPP_QUAD_1032>(acc);
            coef = acc;
            if (!POOL && h == 0 && c < n_mfcc) args.out[f * n_mfcc + c] = acc;
        }
        wave_lds_sync();

        if (POOL) {
            // scrubjay_infer.c:36-66: float64 sums in frame order (the clip's frames are consecutive on this wave)
#pragma clang fp contract(off)
            const int c = lane >> 1;
            const double cv = (double)coef;
            pool_s = pool_s + cv;
            pool_q = pool_q + cv * cv;
            ++pool_t;
            if (last_of_chunk) {
                // the model's fields and output pointers reloaded here through the kernarg segment, not held in SGPRs across the
                // frame loop (mfcc_kernels.hip, pool_finish: 29 of this kernel's SGPRs were spilled)
                const Mfcc512Args *ap = kernarg_of_mfcc512();
                asm volatile("" : "+s"(ap));
                const PoolSvmArgs &pool = ap->pool;
                const SvmModelDev &sm = pool.svm;
                const long clip_o = ap->spans ? ap->spans[clip_f].orig : clip_f;      // ragged batches run in the host's order; results go to the caller's index
                if ((lane & 1) == 0 && c < n_mfcc) {
                    const double mean = pool_s / (double)pool_t;
                    const double var = pool_q / (double)pool_t - mean * mean;
                    const float f_mean = (float)mean, f_std = sqrtf((float)(var > 0 ? var : 0));
                    if (pool.feat) {
                        pool.feat[clip_o * 2L * n_mfcc + c] = f_mean;
                        pool.feat[clip_o * 2L * n_mfcc + n_mfcc + c] = f_std;
                    }
                    feat[c] = (f_mean - scaler[c]) * scaler[64 + c];
                    feat[n_mfcc + c] = (f_std - scaler[n_mfcc + c]) * scaler[64 + n_mfcc + c];
                }
                wave_lds_sync();
                float term = 0.0f;                               // same arithmetic as svm_kernel (svm_kernels.hip)
                for (int sidx = lane; sidx < sm.n_sv; sidx += 64) {
                    const float *sv = sm.sv + (long)sidx * sm.n_features;
                    const float cf = sm.coef[sidx];
                    float d2 = 0.0f;
                    int j = 0;
                    for (; j + 8 <= sm.n_features; j += 8) {     // eight loads in flight, then the same additions in the same order
                        float v[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = sv[j + k];
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const float dd = feat[j + k] - v[k];
                            d2 = d2 + dd * dd;
                        }
                    }
                    for (; j < sm.n_features; ++j) {
                        const float dd = feat[j] - sv[j];
                        d2 = d2 + dd * dd;
                    }
                    term = term + cf * expf(-sm.gamma * d2);
                }
                for (int o = 32; o > 0; o >>= 1) term += __shfl_xor(term, o);
                if (lane == 0) {
                    const float score = term + sm.rho;
                    int label;
                    float p1;
                    svm_binary_tail(score, sm.prob_a, sm.prob_b, label, p1);
                    pool.labels[clip_o] = label;
                    if (pool.decision) pool.decision[clip_o] = score;
                    if (pool.prob1) pool.prob1[clip_o] = p1;
                }
                pool_s = 0.0; pool_q = 0.0; pool_t = 0;
                wave_lds_sync();
            }
        }
    }
}

static size_t lds_bytes_2048(int n_mels, bool pool, bool aub) { return (size_t)q_dct(aub) + (size_t)((n_mels + 1) / 2) * 256 + (pool ? 128 * 4 : 0); }

hipError_t launch_mfcc2048(const Mfcc512Args &args, const GenTables2048 *tables, int blocks, hipStream_t stream, bool pool)
{
    const bool clips = args.frames_per_clip > 0;
    const dim3 g(blocks), b(256);
    const bool aub = args.spectrum != 0 || args.log_mode == 2 || args.stream_framing != 0;
    const size_t Q_BLOCK_BYTES = lds_bytes_2048(args.n_mels, pool, aub);
    if (args.stream_framing && (!clips || (args.samples_per_clip <= 0 && !(pool && args.spans)) || args.hop > args.frame_len)) return hipErrorInvalidConfiguration;      // (ragged: lengths in the spans)
    if (args.in_kind != 0 && (!aub || !clips || args.in_kind < 0 || args.in_kind > 3)) return hipErrorInvalidConfiguration;      // int16: the scrubjay_infer.c front end, clips
    if (pool) {
        if (!clips || args.chunk != args.frames_per_clip || !args.pool.labels || args.pool.svm.n_features != 2 * args.n_mfcc ||
            args.pool.svm.n_features > 64 || (args.log_mode != 0 && args.log_mode != 2))
            return hipErrorInvalidConfiguration;
        if (aub && args.in_kind == 1) hipLaunchKernelGGL((mfcc2048_kernel<true, true, true, 1>), g, b, Q_BLOCK_BYTES, stream, args, tables);
        else if (aub && args.in_kind == 2) hipLaunchKernelGGL((mfcc2048_kernel<true, true, true, 2>), g, b, Q_BLOCK_BYTES, stream, args, tables);
        else if (aub && args.in_kind == 3) hipLaunchKernelGGL((mfcc2048_kernel<true, true, true, 3>), g, b, Q_BLOCK_BYTES, stream, args, tables);
        else if (aub) hipLaunchKernelGGL((mfcc2048_kernel<true, true, true>), g, b, Q_BLOCK_BYTES, stream, args, tables);
        else hipLaunchKernelGGL((mfcc2048_kernel<true, true>), g, b, Q_BLOCK_BYTES, stream, args, tables);
    } else if (clips) {
        if (aub && args.in_kind == 1) hipLaunchKernelGGL((mfcc2048_kernel<true, false, true, 1>), g, b, Q_BLOCK_BYTES, stream, args, tables);
        else if (aub && args.in_kind == 2) hipLaunchKernelGGL((mfcc2048_kernel<true, false, true, 2>), g, b, Q_BLOCK_BYTES, stream, args, tables);
        else if (aub && args.in_kind == 3) hipLaunchKernelGGL((mfcc2048_kernel<true, false, true, 3>), g, b, Q_BLOCK_BYTES, stream, args, tables);
        else if (aub) hipLaunchKernelGGL((mfcc2048_kernel<true, false, true>), g, b, Q_BLOCK_BYTES, stream, args, tables);
        else hipLaunchKernelGGL((mfcc2048_kernel<true, false>), g, b, Q_BLOCK_BYTES, stream, args, tables);
    } else {
        if (aub) hipLaunchKernelGGL((mfcc2048_kernel<false, false, true>), g, b, Q_BLOCK_BYTES, stream, args, tables);
        else hipLaunchKernelGGL((mfcc2048_kernel<false, false>), g, b, Q_BLOCK_BYTES, stream, args, tables);
    }
    return hipGetLastError();
}

int mfcc2048_blocks_per_cu(int n_mels, bool pool, bool aub)
{
    int n = 0;
    hipError_t e;
    if (aub) e = pool ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, mfcc2048_kernel<true, true, true>, 256, lds_bytes_2048(n_mels, true, true))
                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, mfcc2048_kernel<false, false, true>, 256, lds_bytes_2048(n_mels, false, true));
    else e = pool ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, mfcc2048_kernel<true, true>, 256, lds_bytes_2048(n_mels, true, false))
                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, mfcc2048_kernel<false, false>, 256, lds_bytes_2048(n_mels, false, false));
    return e == hipSuccess && n > 0 ? n : 2;
}

}  // namespace dsp
