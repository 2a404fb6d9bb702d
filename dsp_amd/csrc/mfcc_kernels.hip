// mfcc_kernels.hip -- gfx950 (MI355X, CDNA4) MFCC kernels.
//
// One 64-lane wavefront owns one 512-sample audio frame from the HBM load to the
// 13 cepstral coefficients; nothing but the input frame and the coefficients
// touches HBM.  Per frame (reference chain: 2fa/audio/word/c/mfcc.c:142-221):
//
//   load      4 x global_load_dwordx2 per lane: z[l+64a] = x[2n] + i x[2n+1]
//   window    8 v_mul (window pre-scaled by 1/2, tables.cpp)           mfcc.c:142-144
//   FFT       512-point real FFT as a 256-point complex radix-4 DIF:    mfcc.c:16-95
//             every butterfly stage in registers; inter-stage exchanges:
//             1: v_permlane32/16_swap (lane bits 5:4); 2 and 3 (lane bits 3:2,
//             1:0): a 2 KiB per-wave LDS tile with XOR swizzles (bank-conflict
//             free; the last one also restores natural bin order).
//             DSP_X2_LDS / DSP_X3_LDS = 0 switch 2 / 3 to DPP register moves
//             (measured slower: DPP and v_cndmask issue at ~half rate, see
//             DESIGN.md; dataflow model of all forms: tools/emulate_wave_fft.py)
//   untangle  conjugate-pair split with ds_bpermute from the partner lane (packed real FFT)
//   power     |X[k]|^2, k = 0..256                                     mfcc.c:151-155
//   mel       sparse HTK triangles: <= 12 bins per lane + 3-way gather  mfcc.c:158-164
//   log       per-frame ref = max, amin, top_db (v_log_f32)            mfcc.c:169-206
//   DCT-II    default (TILE): once per 16-frame tile as v_mfma_f32_16x16x4_f32,          mfcc.c:210-216
//             D[coef][frame] += A[coef][mel] B[mel][frame] in full fp32 -- the one dense
//             product the chain contains; per-frame form (clip-global log mode): 4 lanes
//             per coefficient + quad DPP reduce
//   store     n_mfcc floats per frame                                  mfcc.c:219-221
//
// The FFT and the 494-non-zero mel product stay on the VALU: they are not dense
// contractions.  The bound is HBM (2048 B in + 52 B out per frame).
#include "diag_guard.hpp"
#include <hip/hip_runtime.h>

#include <algorithm>

#include "mfcc_device.hpp"

// Timing-only diagnostic builds (never shipped), -DDSP_DIAG_MODE=<bit mask>:
//   1 loads + store only, no arithmetic      2 arithmetic only, loads just the first frame
// Outputs are wrong in all of them; only the time is read.
#ifndef DSP_DIAG_MODE
#define DSP_DIAG_MODE 0
#endif
// exchange 1 (slot <-> lane bits 5:4): 1 = through the LDS tile, 0 = v_permlane32/16_swap
#ifndef DSP_X1_LDS
#define DSP_X1_LDS 0
#endif
// conjugate partner of the untangling step: 1 = one b128 LDS round trip, 0 = 4 ds_bpermute_b32
#ifndef DSP_UNT_LDS
#define DSP_UNT_LDS 0
#endif
// frames in flight per wave (software prefetch ring); each costs 8 VGPRs
#ifndef DSP_PREFETCH
#define DSP_PREFETCH 2
#endif
// Wave priority (s_setprio 0..3) rises as a frame progresses: window + refill and the first two butterflies 0, third butterfly
// + exchange 3 at 1, last butterfly .. power at 2, mel + tile at 3.  The SIMD's arbiter then serves the wave closest to
// finishing its frame first ("finish what you started"): fewer frames sit half-done in the LDS queues at any time.  Measured
// on MI355X (tools/ab.py, interleaved, 1 M frames): 0.446 -> 0.408 ms; equal priorities per phase in any other order gain
// less (FFT high 0.424, mel + tile high 0.434, falling priorities 0.421).  Results are bit-identical (scheduling only).
#ifndef DSP_PRIO_LOAD
#define DSP_PRIO_LOAD 0
#endif
#ifndef DSP_PRIO_FFT
#define DSP_PRIO_FFT 0
#endif
#ifndef DSP_PRIO_FFT2
#define DSP_PRIO_FFT2 1
#endif
#ifndef DSP_PRIO_UNT
#define DSP_PRIO_UNT 2
#endif
#ifndef DSP_PRIO_MEL
#define DSP_PRIO_MEL 3
#endif
#ifndef DSP_PRIO_TILE
#define DSP_PRIO_TILE 3
#endif
#define DSP_SETPRIO(from, to) do { if ((from) != (to)) __builtin_amdgcn_s_setprio(to); } while (0)

namespace dsp {

// Diagnostic build only (-DDSP_PF_STAMPS, never shipped): s_memtime at the stations of the fused kernel's clip tail (pool_finish),
// written by lane 0 of the first 1024 waves for their last clip; tools/pf_stamps.py prints the medians.
#ifdef DSP_PF_STAMPS
__device__ unsigned long long g_pf_stamps[8 * 1024];
#define PF_STAMP(k) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 256) g_pf_stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
hipError_t read_pf_stamps(unsigned long long *host, int count) { return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pf_stamps), sizeof(unsigned long long) * (size_t)count); }
#else
#define PF_STAMP(k) do { } while (0)
#endif

namespace {

// IN: element type of the input in HBM (SURVEY.md 8f-1, PCM16 ingestion on device):
//   0 float32 PCM in [-1,1]
//   1 int16 mono                     x = s / 32768                (main_test.c:198-203)
//   2 int16 stereo, channel 0        x = L / 32768                (classifier.c:292-297)
//   3 int16 stereo, channel average  x = 0.5 (L/32768 + R/32768)  (main_test.c:205-217)
// The power-of-two scale is folded into the window (exact), so a lane only converts int -> float.
// `off` is the frame's first sample (per channel).
// STREAM: frames lie back to back and every sample is read exactly once -> nontemporal loads (nothing is displaced in L2).
// Clip mode reads every sample 2.5 times (frames of 400 every 160): those loads must stay cacheable, or the re-reads go
// out to the fabric again (measured with nontemporal loads: 1.96 x the algorithmic bytes; FETCH_SIZE, profiles/r02b).
template <typename V>
__device__ __forceinline__ V ld_frame(const V *p, bool stream) { return stream ? __builtin_nontemporal_load(p) : *p; }

template <int FLEN, int IN, bool STREAM>
__device__ __forceinline__ void load_frame(const void *__restrict__ base, long off, int lane, int frame_len, c32 (&z)[4])
{
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int i = 2 * (lane + 64 * a);
        // FLEN: frame length known at compile time (512: no predicate at all; 400: only lanes 0-7 of a = 3 are live) or 0 = run time
        const int flen = FLEN ? FLEN : frame_len;
        const bool both = i + 1 < flen, one = i < flen;
        if (IN == 0) {
            const float *src = static_cast<const float *>(base) + off;
            if (both) {
                const f2v v = ld_frame(reinterpret_cast<const f2v *>(src + i), STREAM);
                z[a] = {v.x, v.y};
            } else if (one) {
                z[a] = {src[i], 0.0f};
            } else {
                z[a] = {0.0f, 0.0f};
            }
        } else if (IN == 1) {
            // mono int16: the prefetch ring keeps the raw sample pair (one dword, in .x) and converts it when the frame
            // is consumed (unpack_pcm16): 4 VGPRs per frame in flight instead of 8
            const short *src = static_cast<const short *>(base) + off;
            int v = 0;
            if (both) v = ld_frame(reinterpret_cast<const int *>(src + i), STREAM);                  // samples i, i+1
            else if (one) v = (int)(unsigned short)src[i];
            z[a] = {__int_as_float(v), 0.0f};
        } else {
            // interleaved stereo: the ring keeps the two raw L|R dwords, unpack_pcm16 converts at consumption
            const short *src = static_cast<const short *>(base) + 2 * off;
            i2v v = {0, 0};
            if (both) v = ld_frame(reinterpret_cast<const i2v *>(src + 2 * i), STREAM);               // L0 R0 | L1 R1
            else if (one) v.x = *reinterpret_cast<const int *>(src + 2 * i);
            z[a] = {__int_as_float(v.x), __int_as_float(v.y)};
        }
    }
}

// raw ring entry -> the two samples of the complex point (scale folded into the window)
template <int IN>
__device__ __forceinline__ c32 unpack_pcm16(c32 raw)
{
    const int a = __float_as_int(raw.x), b = __float_as_int(raw.y);
    if (IN == 1) return {(float)(short)(a & 0xFFFF), (float)(a >> 16)};                                   // samples i, i+1
    if (IN == 2) return {(float)(short)(a & 0xFFFF), (float)(short)(b & 0xFFFF)};                          // left channel
    return {(float)((short)(a & 0xFFFF) + (a >> 16)), (float)((short)(b & 0xFFFF) + (b >> 16))};          // L + R
}

}  // namespace

// DCT_SPLIT lanes per coefficient, DCT_LEN log-mel values per lane.  GATHER: partial
// sums per mel filter.  FLEN: frame_length at compile time (512, the reference's 400) or 0 = run time (tail predicate on the loads).
// IN: input element type (see load_frame).  CLIPS: frames overlap inside clips (hop < frame).
// Every frame runs the same instructions whatever its position (results do not depend on
// where a frame sits in the batch).  (Carrying two frames per wave through the pipeline
// together was built and measured no faster: the kernel is issue-bound, not latency-bound.)
//
// TILE = 1 (default for the per-frame log mode): the log and the DCT leave the per-frame
// path.  A frame ends with ONE ds_write of its mel energies into a 16-frame LDS tile
// E[mel][frame]; every 16 frames the wave reads the tile back transposed (lane = (mel%4,
// frame)), takes the per-frame maximum and the logs there (10 v_log per 16 frames instead
// of 16) and runs the DCT as v_mfma_f32_16x16x4_f32: D[coef][frame] += A[coef][mel] B[mel][frame]
// with the log-mels as B straight from registers.  The MFMA pipe is otherwise idle; this is
// not a reshaping of the chain into a GEMM but the one dense 13x40 product it already contains.
// Saves ~45 of ~225 VALU issue slots and 10 LDS dwords per frame.
// POOL (TILE, CLIPS, chunk = frames per clip: one wavefront walks one clip): instead of storing the coefficients the
// tile epilogue pools them per clip and the clip ends with the SVM (PoolSvmArgs) -- BASELINE config 5 in one kernel.
// POOL = 2: the epilogue feeds the stop-word net instead (StopNetArgs): classify_signal (stop_detector.c:12-55) in one kernel.
template <int DCT_SPLIT, int DCT_LEN, int GATHER, int FLEN, int IN, int TILE, bool CLIPS, int POOL = 0>
#ifndef DSP_WAVES_PER_EU
#define DSP_WAVES_PER_EU 4
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DSP_WAVES_PER_EU))) void mfcc512_wave_kernel(const Mfcc512Args args)
{
    static_assert(!POOL || (TILE && CLIPS), "pooling lives in the tile epilogue of the clip-mode kernel");
    constexpr int KS = DCT_SPLIT == 2 ? DCT_LEN / 2 : DCT_LEN;    // MFMA k-steps (4 mel filters each)
    constexpr int CT = DCT_SPLIT == 2 ? 2 : 1;                    // 16-coefficient output tiles
    constexpr int WAVE_BYTES = LDS_WAVE_BYTES + (TILE ? LDS_TILE_BYTES : 0) + (POOL == 1 ? LDS_POOL_BYTES : (POOL == 2 ? LDS_STOP_BYTES : 0));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *wl = smem + wib * WAVE_BYTES;
    float *etile = reinterpret_cast<float *>(wl + LDS_WAVE_BYTES);
    // POOL: the clip's running sums live in LDS between tiles (lane c: sum, sum of squares of coefficient c), not in four
    // VGPRs for the whole kernel: the per-frame loop has no registers to spare at 128
    double *pool_acc = reinterpret_cast<double *>(wl + LDS_WAVE_BYTES + LDS_TILE_BYTES);
    (void)etile; (void)pool_acc;
    float2 *xchg = reinterpret_cast<float2 *>(wl + LDS_XCHG);
    float *pbuf = reinterpret_cast<float *>(wl + LDS_XCHG);
    float *part = reinterpret_cast<float *>(wl + LDS_PART);
    float *lmel = reinterpret_cast<float *>(wl + LDS_LOGMEL);

    const LaneTables512 *__restrict__ T = args.tables;

    // ---- per-lane constants (one coalesced dword per field) -----------------
    float win[8];
    constexpr float in_scale = IN == 0 ? 1.0f : (IN == 3 ? 1.0f / 65536.0f : 1.0f / 32768.0f);
#pragma unroll
    for (int i = 0; i < 8; ++i) win[i] = T->win[i][lane] * in_scale;
    c32 tw1[3], tw2[3], tw3[3], twp[2];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        tw1[q] = {T->tw1[2 * q][lane], T->tw1[2 * q + 1][lane]};
        tw2[q] = {T->tw2[2 * q][lane], T->tw2[2 * q + 1][lane]};
        tw3[q] = {T->tw3[2 * q][lane], T->tw3[2 * q + 1][lane]};
    }
    twp[0] = {T->twp[0][lane], T->twp[1][lane]};
    twp[1] = {T->twp[2][lane], T->twp[3][lane]};
    float melw[kMelChunk];
#pragma unroll
    for (int i = 0; i < kMelChunk; ++i) melw[i] = T->mel_w[i][lane];
    const int mel_k0 = T->mel_k0[lane];
    int gat[GATHER];
#pragma unroll
    for (int g = 0; g < GATHER; ++g) gat[g] = T->mel_src[g][lane];
    float dctw[(DCT_LEN + 3) & ~3];
    // the MFMA A operand stays in registers for the reference's shape (10 VGPRs); larger shapes read it from a block-shared
    // LDS copy (once per 16 frames; a global re-read sat latency-exposed in the rolled k loop) to stay within 128 VGPRs
    constexpr bool A_IN_REGS = CT * KS <= 10 && !POOL;      // POOL: the pooling epilogue needs the ten registers (no spills)
    float dcta[CT][KS];
    float *a_lds = reinterpret_cast<float *>(smem + 4 * WAVE_BYTES);
    (void)a_lds;
    if (TILE && !A_IN_REGS) {
        for (int i = threadIdx.x; i < CT * KS * 64; i += 256) a_lds[i] = (&T->dct_a[0][0][0])[(i / (KS * 64)) * (kDctSteps * 64) + i % (KS * 64)];
        __syncthreads();
    }
    if (TILE && A_IN_REGS) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int s = 0; s < KS; ++s) dcta[ct][s] = T->dct_a[ct][s][lane];
    } else if (!TILE) {
#pragma unroll
        for (int i = 0; i < DCT_LEN; ++i) dctw[i] = T->dct_w[i][lane];
    }
    const int n_mels = args.n_mels, n_mfcc = args.n_mfcc;
    // log-mel parts are stored 16-byte aligned (stride DCT_STRIDE floats) so each DCT lane
    // fetches its DCT_LEN values with ds_read_b128 (+ one b64)
    constexpr int DCT_STRIDE = (DCT_LEN + 3) & ~3;
    const int dct_rd = (lane % DCT_SPLIT) * DCT_STRIDE;
    const int lmel_wr = (lane / DCT_LEN) * DCT_STRIDE + lane % DCT_LEN;
    const int dct_c = lane / DCT_SPLIT;
    const bool dct_store = (lane % DCT_SPLIT == 0) && dct_c < n_mfcc;

    // LDS exchange addresses (float2 units), see tools/emulate_wave_fft.py
    //   A2(beta,p,r=4c+d) = 64p + 16beta + 4(c^p) + d : writer lane (beta,c,d) slot p, reader lane (beta,p,d) slot c
    //   A3(beta,p,o,d)    = 64beta + 16o + 4(d^beta) + p : writer lane (beta,p,d) slot o, reader lane (o,p,beta) slot d
    const int d0 = lane & 3, d1 = (lane >> 2) & 3, d2 = lane >> 4;
    const int w3base = 64 * d2 + 4 * (d0 ^ d2) + d1;        // + 16 o
    const int r3base = 64 * d0 + 16 * d2 + d1;              // + 4 (d ^ beta),  beta = d0
    (void)w3base; (void)r3base;
    const bool bit1 = lane & 2, bit0 = lane & 1;
    (void)bit1; (void)bit0;
    const int kap = T->kappa[lane];                 // this lane ends up with bins 64 t + kap
    const int partner = T->partner[lane] << 2;      // byte index for ds_bpermute
    const bool self_paired = kap == 0;              // bins 0/256, 64/192 and 128 pair inside lane 0

    // zero the slots that are only ever read
    if (lane == 0) part[kZeroSlot] = 0.0f;
    if (lane < 16) lmel[64 + lane] = 0.0f;
    lmel[lane] = 0.0f;
    wave_lds_sync();

    const long wave = (long)blockIdx.x * 4 + wib;
    const long n_waves = (long)gridDim.x * 4;
    const unsigned amin_u = __float_as_uint(args.amin);
    const float neg_top_db = -args.top_db;
    const int frame_len = args.frame_len;
    const long n_frames = args.n_frames;

    // The fused clip -> label kernel (POOL = 1) keeps ONE frame in flight: the frame step is inlined once per ring slot, and with it
    // the clip's whole tail (pooling, SVM, libsvm's iteration); one copy less of that cold code in the loop's body is worth 3.4 %
    // there (4.96 -> 4.79 ms per 125 000 clips), while the frame loop itself does not care (frames 0.4162 / 0.4164 ms, clips +0.3 %).
    constexpr int PF = POOL == 1 ? 1 : DSP_PREFETCH;
    // Software prefetch, PF frames deep.  ONE cursor (`pre`) walks this wave's frames
    // and issues their loads; the frame index (and clip) of each ring buffer waits in `fq`
    // until the frame is consumed, so the consumer side needs no cursor of its own.
    // (the fused clip kernels walk whole clips, uniform or ragged: ClipCursor)
    std::conditional_t<POOL != 0, ClipCursor, WaveCursor<CLIPS>> pre;
    if constexpr (POOL != 0) pre.init(wave, n_waves, args.n_clips, args.frames_per_clip, args.hop, args.clip_stride, args.spans, args.samples_per_clip);
    else pre.init(wave, n_waves, args.chunk, n_frames, args.frames_per_clip, CLIPS ? args.hop : frame_len, args.clip_stride);
    c32 ring[PF][4] = {};
    long fq[PF], cq[PF];
    bool lastq[PF];           // POOL: the frame closes its chunk (= its clip)
    // The loads are UNCONDITIONAL (an exhausted cursor requests its last frame again; what comes back is never used): under
    // `if (pre.valid())` the ring slot was a merge of "loaded" and "kept", which the compiler resolved with eight register copies per
    // slot behind s_waitcnt at the loop's exits (32 fewer v_mov in the kernel's text; the steady-state path did not run them:
    // SQ_INSTS_VALU per frame 178.8 -> 178.3, time unchanged -- kept for the shorter code).
    long safe_off = 0;
    auto refill = [&](int d) {
        const bool live = pre.valid();
        if (live) safe_off = pre.off;
        load_frame<FLEN, IN, !CLIPS>(args.in, safe_off, lane, frame_len, ring[d]);
        fq[d] = live ? pre.f : -1; cq[d] = live ? pre.clip : 0; lastq[d] = live && (pre.left == 0 || pre.remaining == 1);
        if (live) pre.next();
    };
#pragma unroll
    for (int d = 0; d < PF; ++d) refill(d);
    // a wave without work leaves -- but in the fused clip kernels only after the block's shared tables below are filled: every thread of
    // the block writes its share of them (a wave that left before that left holes: harmless while the tables were at most 64 entries
    // and the idle waves a block's LAST ones, wrong once ragged batches hand the clips out by counter)
    if (POOL == 0 && fq[0] < 0) return;

    // ---- POOL: per-clip running sums of lane c's coefficient (float64, frames in order), the clip's end -------------
    int pool_t = 0;
    if (POOL == 1 && lane < 32) { pool_acc[2 * lane] = 0.0; pool_acc[2 * lane + 1] = 0.0; }
    // POOL = 2: layers 2-4 of the net (<= 16 x 16 weights + 16 biases each) in a block-shared LDS copy: the clip's tail runs on one
    // lane, and a dependent chain of global loads there would cost tens of microseconds per clip
    float *stop_small = reinterpret_cast<float *>(smem + 4 * WAVE_BYTES) + (TILE && !A_IN_REGS ? CT * KS * 64 : 0);
    (void)stop_small;
    // POOL = 1: the Scaler's offset | scale and the SVM's coefficients in a block-shared LDS copy (the clip's tail read them with
    // dependent global loads, each behind an s_waitcnt vmcnt(0) that also drained the frame prefetch ring: the fused kernel lost
    // 13 % to pool_finish, of which the arithmetic -- support vectors and libsvm's tail -- was a quarter)
    float *svm_small = stop_small;
    (void)svm_small;
    if (POOL == 1) {
        const SvmModelDev &m = args.pool.svm;
        for (int i = threadIdx.x; i < m.n_features; i += 256) { svm_small[i] = m.offset[i]; svm_small[m.n_features + i] = m.scale[i]; }
        for (int i = threadIdx.x; i < m.n_sv; i += 256) svm_small[2 * m.n_features + i] = m.coef[i];
        __syncthreads();
    }
    if (POOL == 2) {
#pragma unroll
        for (int j = 0; j < kStopFusedUnits; ++j) pool_acc[kStopFusedUnits * lane + j] = 0.0;
        const StopModelDev &m = args.stop.m;
        int n_in = m.units[0], off = 0;
        for (int l = 1; l < 4; ++l) {                          // packed: weights of layer l, then its biases
            const int n_w = n_in * m.units[l];
            for (int i = threadIdx.x; i < n_w; i += 256) stop_small[off + i] = m.kernel[l][i];
            for (int i = threadIdx.x; i < m.units[l]; i += 256) stop_small[off + n_w + i] = m.bias[l][i];
            off += n_w + m.units[l];
            n_in = m.units[l];
        }
        __syncthreads();
    }
    if (POOL != 0 && fq[0] < 0) return;
    auto pool_tile = [&](const f4v (&d)[CT], int count) {
        // coefficients of the tile -> LDS as Dt[c][n] (the mel-energy tile has been consumed), then lane c adds its
        // row frame by frame: the same order of float64 additions as mfcc_stats (svm_kernels.hip / scrubjay_infer.c:36-66)
        int lane = threadIdx.x & 63;                         // opaque copy: addresses formed here, not hoisted (see pool_finish)
        asm volatile("" : "+v"(lane));
        const int n = lane & 15, q = lane >> 4;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) etile[(16 * ct + 4 * q + j) * 16 + n] = d[ct][j];
        wave_lds_sync();
#ifdef DSP_DIAG_NO_POOL     // timing-only probe: what the float64 pooling of a tile costs
        if (count < 0)
#endif
        if (lane < n_mfcc) {
#pragma clang fp contract(off)
            double pool_s = pool_acc[2 * lane], pool_q = pool_acc[2 * lane + 1];
            for (int t = 0; t < count; ++t) {
                const double v = (double)etile[lane * 16 + t];
                pool_s = pool_s + v;
                pool_q = pool_q + v * v;
            }
            pool_acc[2 * lane] = pool_s; pool_acc[2 * lane + 1] = pool_q;
        }
        pool_t += count;
        wave_lds_sync();
    };
    // (Out of line as a __noinline__ function -- to take its 44 SGPRs, 9 of them spilled, and 1 300 cold instructions out of the
    // frame loop -- the call's register saves went to scratch INSIDE the loop: 62 VGPRs spilled, 30 ms instead of 5.1.  It stays inline.)
    auto pool_finish = [&](long clip) {
#pragma clang fp contract(off)
        // the kernel arguments through an opaque pointer: the model's fields and the output pointers are (re)loaded HERE, once per
        // clip -- hoisted to the kernel's entry they sat in 44 SGPRs across the frame loop, 9 of them spilled to VGPR lanes
        // (the kernarg segment itself: taking &args would make the compiler copy the argument struct to scratch)
        const Mfcc512Args *ap = kernarg_of_mfcc512();
        asm volatile("" : "+s"(ap));
        if (ap->spans) clip = ap->spans[clip].orig;              // ragged batches run in the host's order; results go to the caller's index
        const PoolSvmArgs &pool = ap->pool;
        const SvmModelDev &m = pool.svm;
        PF_STAMP(0);
        float *z = etile;                                        // 2 * n_mfcc standardised features
        // once per clip: every per-lane address below is formed HERE from an opaque copy of the lane number, so that the
        // compiler cannot hoist a dozen 64-bit addresses out of the frame loop and then spill them (it did: 12 VGPR spills)
        int lane = threadIdx.x & 63;
        asm volatile("" : "+v"(lane));
        if (lane < n_mfcc) {
            const double pool_s = pool_acc[2 * lane], pool_q = pool_acc[2 * lane + 1];
            pool_acc[2 * lane] = 0.0; pool_acc[2 * lane + 1] = 0.0;
            const double mean = pool_s / (double)pool_t;
            const double var = pool_q / (double)pool_t - mean * mean;
            const float f_mean = (float)mean, f_std = sqrtf((float)(var > 0 ? var : 0));
            const float *off_l = svm_small, *scl_l = svm_small + m.n_features;
            z[lane] = (f_mean - off_l[lane]) * scl_l[lane];
            z[n_mfcc + lane] = (f_std - off_l[n_mfcc + lane]) * scl_l[n_mfcc + lane];
            if (pool.feat) {                                     // stores last: nothing below waits for them
                pool.feat[clip * 2L * n_mfcc + lane] = f_mean;
                pool.feat[clip * 2L * n_mfcc + n_mfcc + lane] = f_std;
            }
        }
        wave_lds_sync();
        PF_STAMP(1);
        float term = 0.0f;                                       // same arithmetic as svm_kernel (svm_kernels.hip)
#ifdef DSP_DIAG_NO_SVM      // timing-only probe: what the per-clip support-vector loop costs
        if (m.n_sv < 0)
#endif
        for (int sidx = lane; sidx < m.n_sv; sidx += 64) {
            const float *sv = m.sv + (long)sidx * m.n_features;
            float d2 = 0.0f;
            int j = 0;
            for (; j + 8 <= m.n_features; j += 8) {              // eight loads in flight, then the same additions in the same order
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = sv[j + k];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float dd = z[j + k] - v[k];
                    d2 = d2 + dd * dd;
                }
            }
            for (; j < m.n_features; ++j) {
                const float dd = z[j] - sv[j];
                d2 = d2 + dd * dd;
            }
            term = term + svm_small[2 * m.n_features + sidx] * expf(-m.gamma * d2);
        }
        PF_STAMP(2);
        for (int o = 32; o > 0; o >>= 1) term += __shfl_xor(term, o);
        PF_STAMP(3);
        if (lane == 0) {
            const float score = term + m.rho;
            int label;
            float p1;
#ifdef DSP_DIAG_NO_SVMTAIL      // timing-only probe: libsvm's label / probability tail on lane 0
            label = score > 0; p1 = score;
#else
            svm_binary_tail(score, m.prob_a, m.prob_b, label, p1);
#endif
            pool.labels[clip] = label;
            if (pool.decision) pool.decision[clip] = score;
            if (pool.prob1) pool.prob1[clip] = p1;
        }
        PF_STAMP(4);
        wave_lds_sync();
        PF_STAMP(5);
        pool_t = 0;
    };
    // ---- POOL = 2: the stop-word net's first layer accumulated tile by tile (audio_classifier_inference.c:18-36, 44-47) ----
    // lane (c = lane % 16, tq = lane / 16) takes coefficient c of the tile's frames 4 tq .. 4 tq + 3 -- input index
    // i = c * max_frames + t, the coefficient-major view of stop_detector.c:36-50.  The scaler is folded into the weights at model
    // creation, w (x - mean) / div = x A + B (StopModelDev::fold_a, pad_b): ONE 16-byte load and four float64 FMAs per input, no
    // division in the loop; the B terms of the live inputs are a per-T constant added at the end.  float64 sums: their order
    // shows at 1e-16; the folding moves a term by <= 1e-7 relative (the reference itself adds 6500 fp32 terms in sequence).
    auto stop_tile = [&](const f4v (&d)[CT], int count) {
        const Mfcc512Args *ap = kernarg_of_mfcc512();          // the model's fields reloaded here, not held in SGPRs across the frame loop (pool_finish)
        asm volatile("" : "+s"(ap));
        const StopModelDev &m = ap->stop.m;
        int lane = threadIdx.x & 63;
        asm volatile("" : "+v"(lane));
        const int n = lane & 15, q = lane >> 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) etile[(4 * q + j) * 16 + n] = d[0][j];       // Dt[c][n], one coefficient tile
        wave_lds_sync();
        const int c = lane & 15, tq = lane >> 4;
        const bool live_c = c < m.n_coef;
        const f4v *A = reinterpret_cast<const f4v *>(m.fold_a);
        double acc[kStopFusedUnits];
#pragma unroll
        for (int j = 0; j < kStopFusedUnits; ++j) acc[j] = pool_acc[kStopFusedUnits * lane + j];
#pragma unroll 1
        for (int k0 = 0; k0 < 4; k0 += 2) {                                    // two inputs at a time: their loads go out together; four at once spill
            f4v a[2];
            float x[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int tt = 4 * tq + k0 + k, t = pool_t + tt;
                const bool on = live_c && tt < count && t < m.max_frames;      // stop_detector.c:26-30: frames past max_frames are dropped
                a[k] = A[on ? c * m.max_frames + t : 0];
                x[k] = on ? etile[c * 16 + tt] : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const double xd = (double)x[k];
#pragma unroll
                for (int j = 0; j < kStopFusedUnits; ++j) acc[j] = fma((double)a[k][j], xd, acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < kStopFusedUnits; ++j) pool_acc[kStopFusedUnits * lane + j] = acc[j];
        pool_t += count;
        wave_lds_sync();
    };
    auto stop_finish = [&](long clip) {
#pragma clang fp contract(off)
        const Mfcc512Args *ap = kernarg_of_mfcc512();
        asm volatile("" : "+s"(ap));
        if (ap->spans) clip = ap->spans[clip].orig;              // ragged batches run in the host's order; results go to the caller's index
        const StopModelDev &m = ap->stop.m;
        int lane = threadIdx.x & 63;
        asm volatile("" : "+v"(lane));
        double acc[kStopFusedUnits];
#pragma unroll
        for (int j = 0; j < kStopFusedUnits; ++j) {
            acc[j] = pool_acc[kStopFusedUnits * lane + j];
            pool_acc[kStopFusedUnits * lane + j] = 0.0;
            for (int o = 32; o > 0; o >>= 1) acc[j] += __shfl_xor(acc[j], o);
        }
        if (lane == 0) {
            // layers 1 (bias + zero-padded frames + live sums) .. 4 and the sigmoid, as stop_tail_kernel
            const int u1 = m.units[0];
            const int Tc = pool_t < m.max_frames ? pool_t : m.max_frames;
            float *h = etile;                                                    // two rows of kStopMaxUnits activations (the tile is free here)
#pragma unroll
            for (int j = 0; j < kStopFusedUnits; ++j)
                if (j < u1) {
                    const float sum = (float)((double)m.bias[0][j] + m.pad_b[(long)Tc * u1 + j] + acc[j]);
                    h[j] = sum > 0.0f ? sum : 0.0f;
                }
            int n_in = u1, off = 0;
            for (int l = 1; l < 4; ++l) {                                        // dense_forward, :18-35
                const int n_out = m.units[l];
                const float *src = h + ((l - 1) & 1) * kStopMaxUnits;
                float *dst = h + (l & 1) * kStopMaxUnits;
                const float *wl_ = stop_small + off, *bl_ = wl_ + n_in * n_out;
                off += n_in * n_out + n_out;
                for (int j = 0; j < n_out; ++j) {
                    float sum = bl_[j];
                    for (int i = 0; i < n_in; ++i) sum = sum + wl_[i * n_out + j] * src[i];
                    dst[j] = (l < 3 && !(sum > 0.0f)) ? 0.0f : sum;
                }
                n_in = n_out;
            }
            ap->stop.prob[clip] = 1.0f / (1.0f + expf(-h[kStopMaxUnits]));     // :13-15 (layer 4's output sits in row 1)
        }
        pool_t = 0;
        wave_lds_sync();
    };
    (void)pool_tile; (void)pool_finish; (void)pool_t; (void)stop_tile; (void)stop_finish;

    // ---- 16-frame tile epilogue (TILE): log + DCT for the frames in slots [0, count) ----
    int slot = 0;               // frames in the tile
    long fb0 = 0, fb1 = 0;      // first frame of slots 0..7 / 8..15 (each half is 8 consecutive frames)
    auto flush = [&](int count) {
        wave_lds_sync();
        int lane = threadIdx.x & 63;                         // POOL: opaque copy, the epilogue's addresses are formed here (see pool_finish)
        if (POOL) asm volatile("" : "+v"(lane));
        const int n = lane & 15, q = lane >> 4;
        // reference shape: the tile column stays in registers between the max and the log passes; larger shapes
        // (more k-steps or two coefficient tiles) re-read it from LDS in a rolled loop instead of spilling
        float ev[KS];
        unsigned mx = amin_u;
        if (A_IN_REGS) {
#pragma unroll
            for (int s = 0; s < KS; ++s) ev[s] = etile[64 * s + 16 * q + (n ^ s)];     // mel 4s+q of frame n
#pragma unroll
            for (int s = 0; s < KS; ++s) mx = max(mx, __float_as_uint(ev[s]));
        } else {
#pragma unroll 2
            for (int s = 0; s < KS; ++s) mx = max(mx, __float_as_uint(etile[64 * s + 16 * q + (n ^ s)]));
        }
        {   // max over the four 16-lane rows (same frame, other mel residues)
            auto r = __builtin_amdgcn_permlane16_swap(mx, mx, false, false);
            mx = max(r[0], r[1]);
            r = __builtin_amdgcn_permlane32_swap(mx, mx, false, false);
            mx = max(r[0], r[1]);
        }
        const float rinv = __builtin_amdgcn_rcpf(__uint_as_float(mx));
        const float k10 = 3.01029995663981195f;            // 10 * log10(2)
        f4v acc[CT][2];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct][0] = acc[ct][1] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
        auto kstep = [&](int s, float e_s, int par) {
            const float ec = __uint_as_float(max(__float_as_uint(e_s), amin_u));
            float db = k10 * __builtin_amdgcn_logf(ec * rinv);
            db = __builtin_amdgcn_fmed3f(db, neg_top_db, 0.0f);
            if (4 * s + q >= n_mels) db = 0.0f;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const float a = A_IN_REGS ? dcta[ct][s < KS ? s : 0] : a_lds[(ct * KS + s) * 64 + lane];
                acc[ct][par] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, db, acc[ct][par], 0, 0, 0);
            }
        };
        if (A_IN_REGS) {
#pragma unroll
            for (int s = 0; s < KS; ++s) kstep(s, ev[s], s & 1);
        } else {
            static_assert(KS % 2 == 0, "two accumulator chains");
#pragma unroll 1
            for (int s = 0; s < KS; s += 2) {
                kstep(s, etile[64 * s + 16 * q + (n ^ s)], 0);
                kstep(s + 1, etile[64 * (s + 1) + 16 * q + (n ^ (s + 1))], 1);
            }
        }
        if (POOL) {
            f4v dd[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) dd[ct] = acc[ct][0] + acc[ct][1];
            wave_lds_sync();                                 // every lane has read its tile column
#ifdef DSP_DIAG_NO_POOLTILE     // timing-only probe: the walk and the tile epilogue of the fused kernel without its pooling
            if (dd[0][0] == 123.456f) args.pool.labels[0] = 1;
            return;
#endif
            if (POOL == 2) { if constexpr (CT == 1) stop_tile(dd, count); }
            else pool_tile(dd, count);
            return;
        }
        const long fl = (n < 8 ? fb0 : fb1 - 8) + n;
        const bool ok = n < count && fl < n_frames;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const f4v d = acc[ct][0] + acc[ct][1];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = 16 * ct + 4 * q + j;
                if (ok && c < n_mfcc) args.out[fl * n_mfcc + c] = d[j];
            }
        }
        wave_lds_sync();
    };
    (void)flush;

    // one frame: ring[d] -> coefficients (or a tile slot); refills ring[d]; false when this was the wave's last frame
    auto step = [&](int rd) -> bool {
        c32 (&nxt)[4] = ring[rd];
        const long f = fq[rd];
        const long clip_f = cq[rd];          // clip of frame f (clip mode)
        const bool last_f = lastq[rd];      // POOL: f closes its clip
        (void)clip_f; (void)last_f;
        c32 s[4];
        DSP_SETPRIO(DSP_PRIO_TILE, DSP_PRIO_LOAD);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const c32 x = IN == 0 ? nxt[a] : unpack_pcm16<IN>(nxt[a]);
            s[a] = {x.x * win[2 * a], x.y * win[2 * a + 1]};
        }
#if !(DSP_DIAG_MODE & 2)
        refill(rd);                         // hidden by the work below
#else
        fq[rd] = pre.valid() ? pre.f : -1; if (pre.valid()) pre.next();
#endif
        const bool more = fq[(rd + 1) % PF] >= 0;
#if DSP_DIAG_MODE == 1
        {
            float c = ((s[0].x + s[0].y) + (s[1].x + s[1].y)) + ((s[2].x + s[2].y) + (s[3].x + s[3].y));
            c += dpp<DPP_QUAD_1032>(c);
            if (dct_store) args.out[f * n_mfcc + dct_c] = c;
        }
        return more;
#endif

        // ---- 256-point complex FFT, radix-4 DIF --------------------------------
        DSP_SETPRIO(DSP_PRIO_LOAD, DSP_PRIO_FFT);
        radix4(s);                                             // digit a (bits 7:6)
#pragma unroll
        for (int q = 1; q < 4; ++q) s[q] = cmul(s[q], tw1[q - 1]);
#if DSP_X1_LDS
        // exchange 1: slot a <-> lane bits 5:4 through LDS.  A1 = 64 a + ((16 h + r) ^ 16 (a & 1)):
        // writer lane (h, r) slot a, reader lane (h, r) slot a' takes lane (a', r) slot h
#pragma unroll
        for (int a = 0; a < 4; ++a) xchg[64 * a + (lane ^ (16 * (a & 1)))] = {s[a].x, s[a].y};
        wave_lds_sync();
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float2 v = xchg[64 * d2 + ((16 * a + (lane & 15)) ^ (16 * (d2 & 1)))];
            s[a] = {v.x, v.y};
        }
        wave_lds_sync();
#else
        // exchange 1: slot <-> lane bits 5:4, in registers
        swap_hi32(s[0].x, s[2].x); swap_hi32(s[0].y, s[2].y);
        swap_hi32(s[1].x, s[3].x); swap_hi32(s[1].y, s[3].y);
        swap_odd16(s[0].x, s[1].x); swap_odd16(s[0].y, s[1].y);
        swap_odd16(s[2].x, s[3].x); swap_odd16(s[2].y, s[3].y);
#endif
        radix4(s);                                             // digit b (bits 5:4)
#pragma unroll
        for (int q = 1; q < 4; ++q) s[q] = cmul(s[q], tw2[q - 1]);
#if DSP_X2_LDS
        // exchange 2: slot <-> lane bits 3:2, through LDS
#pragma unroll
        for (int p = 0; p < 4; ++p) xchg[(lane ^ (4 * p)) + 64 * p] = {s[p].x, s[p].y};
        wave_lds_sync();
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float2 v = xchg[64 * d1 + 16 * d2 + 4 * (c ^ d1) + d0];
            s[c] = {v.x, v.y};
        }
        wave_lds_sync();
#else
        // exchange 2: slot <-> lane bits 3:2, DPP row moves
        swap_lane8(s[0].x, s[2].x); swap_lane8(s[0].y, s[2].y);
        swap_lane8(s[1].x, s[3].x); swap_lane8(s[1].y, s[3].y);
        swap_lane4(s[0].x, s[1].x); swap_lane4(s[0].y, s[1].y);
        swap_lane4(s[2].x, s[3].x); swap_lane4(s[2].y, s[3].y);
#endif
        DSP_SETPRIO(DSP_PRIO_FFT, DSP_PRIO_FFT2);
#ifdef DSP_DIAG_SNOPS      // timing-only probe: extra scalar / vector issue slots per frame (is the kernel bound by instruction issue?)
#pragma unroll
        for (int i_ = 0; i_ < DSP_DIAG_SNOPS; ++i_) asm volatile("s_nop 0");
#endif
#ifdef DSP_DIAG_VNOPS
#pragma unroll
        for (int i_ = 0; i_ < DSP_DIAG_VNOPS; ++i_) asm volatile("v_mov_b32 %0, %0" : "+v"(s[i_ & 3].x));
#endif
        radix4(s);                                             // digit c (bits 3:2)
#pragma unroll
        for (int q = 1; q < 4; ++q) s[q] = cmul(s[q], tw3[q - 1]);
#if DSP_X3_LDS
        // exchange 3: slot <-> lane bits 1:0, through LDS; reader lane = k mod 64
#pragma unroll
        for (int o = 0; o < 4; ++o) xchg[w3base + 16 * o] = {s[o].x, s[o].y};
        wave_lds_sync();
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
            const float2 v = xchg[r3base + 4 * (dd ^ d0)];
            s[dd] = {v.x, v.y};
        }
        wave_lds_sync();
#else
        // exchange 3: slot <-> lane bits 1:0, DPP quad permutes
        swap_quad<DPP_QUAD_2301>(s[0].x, s[2].x, bit1); swap_quad<DPP_QUAD_2301>(s[0].y, s[2].y, bit1);
        swap_quad<DPP_QUAD_2301>(s[1].x, s[3].x, bit1); swap_quad<DPP_QUAD_2301>(s[1].y, s[3].y, bit1);
        swap_quad<DPP_QUAD_1032>(s[0].x, s[1].x, bit0); swap_quad<DPP_QUAD_1032>(s[0].y, s[1].y, bit0);
        swap_quad<DPP_QUAD_1032>(s[2].x, s[3].x, bit0); swap_quad<DPP_QUAD_1032>(s[2].y, s[3].y, bit0);
#endif

        // ---- last butterfly, packed-real untangling, power spectrum ----------------
        // the lane with bins kap, kap+64 pairs them with 256-kap and 192-kap; both live
        // in the partner lane (slots 3 and 2).  Lane 0 (kap = 0) pairs inside itself.
        DSP_SETPRIO(DSP_PRIO_FFT2, DSP_PRIO_UNT);
        radix4(s);                                             // digit d: s[t] = Z[64 t + kap] / 2
        c32 b, d;
#if DSP_UNT_LDS
        {
            float4 *pt = reinterpret_cast<float4 *>(xchg);
            pt[lane] = make_float4(s[2].x, s[2].y, s[3].x, s[3].y);
            wave_lds_sync();
            const float4 v = pt[partner >> 2];
            wave_lds_sync();
            d = {v.x, v.y};
            b = {v.z, v.w};
        }
#else
        b.x = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(s[3].x)));
        b.y = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(s[3].y)));
        d.x = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(s[2].x)));
        d.y = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(s[2].y)));
#endif
        // selects, not a branch: lane 0 exists in every wave, so a branch is always taken
        b.x = self_paired ? s[0].x : b.x; b.y = self_paired ? s[0].y : b.y;
        d.x = self_paired ? s[3].x : d.x; d.y = self_paired ? s[3].y : d.y;
        float P[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const c32 x = s[h], v = h == 0 ? b : d;
            const c32 E = {x.x + v.x, x.y - v.y};           // Z[k] + conj(Z[N-k])
            const c32 O = {x.x - v.x, x.y + v.y};           // Z[k] - conj(Z[N-k])
            const c32 Tw = cmul(O, twp[h]);                 // W512^k * O
            const float xr = E.x + Tw.y, xi = E.y - Tw.x;   // X[k]
            const float mr = E.x - Tw.y, mi = E.y + Tw.x;   // X[256-k] (conjugated)
            P[2 * h] = xr * xr + xi * xi;
            P[2 * h + 1] = mr * mr + mi * mi;
        }
        const float p128 = 4.0f * (s[2].x * s[2].x + s[2].y * s[2].y);   // lane 0: |Z[128]|^2 un-halved
        pbuf[kap] = P[0];
        pbuf[64 + kap] = P[2];
        pbuf[192 - kap] = P[3];
        pbuf[256 - kap] = P[1];
        if (self_paired) pbuf[128] = p128;
        DSP_SETPRIO(DSP_PRIO_UNT, DSP_PRIO_MEL);
        wave_lds_sync();

        // ---- sparse mel filterbank -------------------------------------------
        {
            const float *rd = pbuf + mel_k0;
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < kMelChunk; ++i) acc = fmaf(melw[i], rd[i], acc);
            part[lane] = acc;
        }
        wave_lds_sync();
        float e = part[gat[0]];
#pragma unroll
        for (int g = 1; g < GATHER; ++g) e += part[gat[g]];
        if (lane >= n_mels) e = 0.0f;

        DSP_SETPRIO(DSP_PRIO_MEL, DSP_PRIO_TILE);
        if (TILE) {
            if ((slot & 7) == 0) { if (slot == 0) fb0 = f; else fb1 = f; }
            etile[16 * lane + (slot ^ (lane >> 2))] = e;
            const bool clip_ends = POOL && last_f;
            if (clip_ends) PF_STAMP(6);
            if (++slot == 16 || !more || clip_ends) { flush(slot); slot = 0; }
#if !defined(DSP_DIAG_NO_POOLTILE) && !defined(DSP_DIAG_NO_POOLFINISH)
            if (clip_ends) { if (POOL == 2) stop_finish(clip_f); else pool_finish(clip_f); }
#endif
            return more;
        }

        // ---- 10 log10 with per-frame reference (mfcc.c:169-206) ---------------
        // 10 log10(max(e,amin)) - 10 log10(ref) evaluated as one log of the ratio: no
        // cancellation between two ~-100 dB terms, and exactly invariant to a
        // power-of-two gain on the input.  (e, amin, ref are non-negative: their max
        // is an unsigned-integer max of the bit patterns, no NaN canonicalisation.)
        const float k10 = 3.01029995663981195f;            // 10 * log10(2)
        if (args.log_mode != 0) {
            // librosa power_to_db(ref = 1.0, top_db over the clip), keyword_classifier.py:42-55
            const float top = __uint_as_float(max(__float_as_uint(wave_max_nonneg(e)), amin_u));
            const float top_db_val = k10 * __builtin_amdgcn_logf(top);      // frame maximum in dB
            if (args.frame_max != nullptr) {        // pass 1
                if (lane == 0) args.frame_max[f] = top_db_val;
                return more;
            }
            const float floor_db = args.clip_floor ? args.clip_floor[clip_f] : top_db_val + neg_top_db;
            const float ec = __uint_as_float(max(__float_as_uint(e), amin_u));
            const float db = fmaxf(k10 * __builtin_amdgcn_logf(ec), floor_db);
            if (lane < n_mels) lmel[lmel_wr] = db;
        } else {
            const float ref = __uint_as_float(max(__float_as_uint(wave_max_nonneg(e)), amin_u));
            const float ec = __uint_as_float(max(__float_as_uint(e), amin_u));
            float db = k10 * __builtin_amdgcn_logf(ec * __builtin_amdgcn_rcpf(ref));
            db = __builtin_amdgcn_fmed3f(db, neg_top_db, 0.0f); // clamp to [-top_db, 0]: the frame max is exactly 0
            if (lane < n_mels) lmel[lmel_wr] = db;
        }
        wave_lds_sync();

        // ---- DCT-II ------------------------------------------------------------
        {
            const float *rd = lmel + dct_rd;
            float c = 0.0f;
#pragma unroll
            for (int i = 0; i + 4 <= DCT_LEN; i += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(rd + i);
                c = fmaf(dctw[i], v.x, c);
                c = fmaf(dctw[i + 1], v.y, c);
                c = fmaf(dctw[i + 2], v.z, c);
                c = fmaf(dctw[i + 3], v.w, c);
            }
            if (DCT_LEN % 4) {
                const float2 v = *reinterpret_cast<const float2 *>(rd + (DCT_LEN & ~3));
                c = fmaf(dctw[DCT_LEN & ~3], v.x, c);
                c = fmaf(dctw[(DCT_LEN & ~3) + 1], v.y, c);
            }
            c += dpp<DPP_QUAD_1032>(c);
            if (DCT_SPLIT == 4) c += dpp<DPP_QUAD_2301>(c);
            if (dct_store) args.out[f * n_mfcc + dct_c] = c;
        }
        wave_lds_sync();
        return more;
    };

    static_assert(DSP_PREFETCH >= 1 && DSP_PREFETCH <= 3, "prefetch ring depth");
    while (true) {
        if (!step(0)) return;
        if (PF > 1 && !step(PF > 1 ? 1 : 0)) return;
        if (PF > 2 && !step(PF > 2 ? 2 : 0)) return;
    }
}

// -----------------------------------------------------------------------------

// Instantiations: (DCT_SPLIT, DCT_LEN, GATHER) x FLEN x TILE x CLIPS for float input; PCM16 input (always clips, tile
// epilogue) for the reference's shape (13 x 40) only.  FLEN = 400 (the reference's framing, -5 % in clip mode) exists for the
// shapes of BASELINE configs 4 and 5; other frame lengths below 512 take the run-time predicate (FLEN = 0).
#define DSP_FOR_SHAPES(X) X(4, 10, 3) X(4, 10, 6) X(4, 16, 3) X(4, 16, 6) X(2, 20, 3) X(2, 20, 6)

// S, L: DCT_SPLIT, DCT_LEN of the instantiation (shapes whose MFMA A operand does not fit registers keep it in LDS)
template <int S, int L>
static size_t lds_bytes(bool tile, int pool = 0, int stop_small_floats = 0)
{
    constexpr int KS = S == 2 ? L / 2 : L, CT = S == 2 ? 2 : 1;
    return (size_t)4 * (LDS_WAVE_BYTES + (tile ? LDS_TILE_BYTES : 0) + (pool == 1 ? LDS_POOL_BYTES : (pool == 2 ? LDS_STOP_BYTES : 0))) +
           (tile && (CT * KS > 10 || pool) ? (size_t)CT * KS * 64 * 4 : 0) + (pool ? (size_t)((stop_small_floats + 3) & ~3) * 4 : 0);      // pool = 1: offset | scale | coef
}

static int stop_small_floats(const StopModelDev &m)
{
    int n = 0;
    for (int l = 1; l < 4; ++l) n += m.units[l - 1] * m.units[l] + m.units[l];
    return n;
}

template <int S, int L, int G>
constexpr bool kHas400 = (S == 4 && L == 10 && G == 3) || (S == 2 && L == 20 && G == 3);

// 512 -> 512, 400 -> 400 where instantiated, anything else -> 0 (run-time predicate)
template <int S, int L, int G>
static int flen_of(int frame_len) { return frame_len == 512 ? 512 : (frame_len == 400 && kHas400<S, L, G> ? 400 : 0); }

template <int S, int L, int G, int FLEN, int IN, int TILE>
static void launch_flen(const Mfcc512Args &args, bool clips, dim3 g, dim3 b, size_t lds, hipStream_t stream)
{
    if (IN == 0 && !clips) hipLaunchKernelGGL((mfcc512_wave_kernel<S, L, G, FLEN, 0, TILE, false>), g, b, lds, stream, args);
    else hipLaunchKernelGGL((mfcc512_wave_kernel<S, L, G, FLEN, IN, TILE, true>), g, b, lds, stream, args);
}

template <int S, int L, int G, int IN, int TILE>
static hipError_t launch_one(const Mfcc512Args &args, bool clips, int blocks, hipStream_t stream)
{
    const size_t lds = lds_bytes<S, L>(TILE);
    const dim3 g(blocks), b(256);
    const int flen = flen_of<S, L, G>(args.frame_len);
    if (flen == 512) launch_flen<S, L, G, 512, IN, TILE>(args, clips, g, b, lds, stream);
    else if (flen == 400) { if constexpr (kHas400<S, L, G>) launch_flen<S, L, G, 400, IN, TILE>(args, clips, g, b, lds, stream); }
    else launch_flen<S, L, G, 0, IN, TILE>(args, clips, g, b, lds, stream);
    return hipGetLastError();
}

hipError_t launch_mfcc512_pool(const Mfcc512Args &args, int dct_split, int dct_len, int gather, int blocks, hipStream_t stream)
{
    if (args.frames_per_clip <= 0 || args.chunk != args.frames_per_clip || args.log_mode != 0 || args.in_kind < 0 || args.in_kind > 3 || !args.pool.labels ||
        args.pool.svm.n_features != 2 * args.n_mfcc || args.pool.svm.n_features > 64 || args.pool.svm.n_sv < 1 || args.pool.svm.n_sv > 2048)
        return hipErrorInvalidConfiguration;
    const dim3 g(blocks), b(256);
    if (args.in_kind != 0) {
        // int16 PCM in the fused clip -> label kernel: the reference framing (frame 400) on the shapes of 13 and 20
        // coefficients of 40 mel energies -- what scrubjay_infer.c's and stop_detector.c's callers decode from their WAV files
#define DSP_LAUNCH_POOL_PCM(S, L, G)                                                                                        \
        if (dct_split == S && dct_len == L && gather == G && args.frame_len == 400) {                                        \
            const size_t lds = lds_bytes<S, L>(true, 1, 2 * args.pool.svm.n_features + args.pool.svm.n_sv);                 \
            if (args.in_kind == 1) hipLaunchKernelGGL((mfcc512_wave_kernel<S, L, G, 400, 1, 1, true, 1>), g, b, lds, stream, args);      \
            else if (args.in_kind == 2) hipLaunchKernelGGL((mfcc512_wave_kernel<S, L, G, 400, 2, 1, true, 1>), g, b, lds, stream, args); \
            else hipLaunchKernelGGL((mfcc512_wave_kernel<S, L, G, 400, 3, 1, true, 1>), g, b, lds, stream, args);          \
            return hipGetLastError();                                                                                       \
        }
        DSP_LAUNCH_POOL_PCM(4, 10, 3)
        DSP_LAUNCH_POOL_PCM(2, 20, 3)
#undef DSP_LAUNCH_POOL_PCM
        return hipErrorInvalidConfiguration;
    }
#define DSP_LAUNCH_POOL(S, L, G)                                                                                            \
    if (dct_split == S && dct_len == L && gather == G) {                                                                    \
        const size_t lds = lds_bytes<S, L>(true, 1, 2 * args.pool.svm.n_features + args.pool.svm.n_sv);                    \
        const int flen = flen_of<S, L, G>(args.frame_len);                                                                  \
        if (flen == 512) hipLaunchKernelGGL((mfcc512_wave_kernel<S, L, G, 512, 0, 1, true, 1>), g, b, lds, stream, args);   \
        else if (flen == 400) {                                                                                             \
            if constexpr (kHas400<S, L, G>) hipLaunchKernelGGL((mfcc512_wave_kernel<S, L, G, 400, 0, 1, true, 1>), g, b, lds, stream, args); \
        } else hipLaunchKernelGGL((mfcc512_wave_kernel<S, L, G, 0, 0, 1, true, 1>), g, b, lds, stream, args);             \
        return hipGetLastError();                                                                                           \
    }
    DSP_FOR_SHAPES(DSP_LAUNCH_POOL)
#undef DSP_LAUNCH_POOL
    return hipErrorInvalidConfiguration;
}

typedef void (*StopKernel)(const Mfcc512Args);
static StopKernel stop_kernel_of(int in_kind, int gather, int frame_len)
{
    if (in_kind == 1) return mfcc512_wave_kernel<4, 10, 3, 400, 1, 1, true, 2>;       // int16 PCM: main_test.c's reader feeds classify_signal
    if (in_kind == 2) return mfcc512_wave_kernel<4, 10, 3, 400, 2, 1, true, 2>;
    if (in_kind == 3) return mfcc512_wave_kernel<4, 10, 3, 400, 3, 1, true, 2>;
    if (gather == 3 && frame_len == 400) return mfcc512_wave_kernel<4, 10, 3, 400, 0, 1, true, 2>;
    if (gather == 3) return mfcc512_wave_kernel<4, 10, 3, 0, 0, 1, true, 2>;
    if (gather == 6) return mfcc512_wave_kernel<4, 10, 6, 0, 0, 1, true, 2>;
    return nullptr;
}

// the blocks launch_mfcc512_stop will start for a caller's count: at most what the chip holds with this variant's LDS (the host lays
// ragged batches out for that many wavefronts)
int mfcc512_stop_grid(int blocks, const StopModelDev &m, int in_kind, int gather, int frame_len)
{
    const StopKernel kernel = stop_kernel_of(in_kind, gather, frame_len);
    if (!kernel) return blocks;
    const size_t lds = lds_bytes<4, 10>(true, 2, stop_small_floats(m));
    int per_cu = 0, dev = 0, n_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n_cu > 0)
        blocks = std::min(blocks, per_cu * n_cu);
    return std::max(blocks, 1);
}

// classify_signal fused (POOL = 2): the reference's shape only (13 coefficients of 40 mel energies: DCT_SPLIT 4, DCT_LEN 10)
hipError_t launch_mfcc512_stop(const Mfcc512Args &args, int dct_split, int dct_len, int gather, int blocks, hipStream_t stream)
{
    const StopModelDev &m = args.stop.m;
    if (args.frames_per_clip <= 0 || args.chunk != args.frames_per_clip || args.log_mode != 0 || args.in_kind < 0 || args.in_kind > 3 || !args.stop.prob ||
        (args.in_kind != 0 && !(gather == 3 && args.frame_len == 400)) || dct_split != 4 || dct_len != 10 || m.n_coef != args.n_mfcc || m.n_coef > 16 || m.units[0] < 1 || m.units[0] > kStopFusedUnits ||
        !m.fold_a || !m.pad_b)
        return hipErrorInvalidConfiguration;
    for (int l = 1; l < 4; ++l)
        if (m.units[l] < 1 || m.units[l] > kStopMaxUnits) return hipErrorInvalidConfiguration;
    const size_t lds = lds_bytes<4, 10>(true, 2, stop_small_floats(m));
    const StopKernel kernel = stop_kernel_of(args.in_kind, gather, args.frame_len);
    if (!kernel) return hipErrorInvalidConfiguration;
    // persistent-style grid: the blocks the chip really holds with this variant's LDS (the caller's count is the plain kernel's)
    blocks = mfcc512_stop_grid(blocks, m, args.in_kind, gather, args.frame_len);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), lds, stream, args);
    return hipGetLastError();
}

hipError_t launch_mfcc512(const Mfcc512Args &args, int dct_split, int dct_len, int gather, int blocks,
                          hipStream_t stream, bool tile)
{
    const bool clips = args.frames_per_clip > 0;
    if (tile && (args.log_mode != 0 || args.chunk % 8 != 0)) return hipErrorInvalidConfiguration;
    if (args.in_kind != 0) {
        if (!(dct_split == 4 && dct_len == 10 && gather == 3) || !tile || !clips) return hipErrorInvalidConfiguration;
        if (args.in_kind == 1) return launch_one<4, 10, 3, 1, 1>(args, true, blocks, stream);
        if (args.in_kind == 2) return launch_one<4, 10, 3, 2, 1>(args, true, blocks, stream);
        if (args.in_kind == 3) return launch_one<4, 10, 3, 3, 1>(args, true, blocks, stream);
        return hipErrorInvalidConfiguration;
    }
#define DSP_LAUNCH(S, L, G)                                                                  \
    if (dct_split == S && dct_len == L && gather == G)                                       \
        return tile ? launch_one<S, L, G, 0, 1>(args, clips, blocks, stream)                 \
                    : launch_one<S, L, G, 0, 0>(args, clips, blocks, stream);
    DSP_FOR_SHAPES(DSP_LAUNCH)
#undef DSP_LAUNCH
    return hipErrorInvalidConfiguration;
}

__global__ void clip_floor_kernel(const float *__restrict__ frame_max, long n_clips, int fpc, float top_db,
                                  float *__restrict__ clip_floor)
{
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_clips) return;
    float m = -INFINITY;
    for (int t = 0; t < fpc; ++t) m = fmaxf(m, frame_max[c * fpc + t]);
    clip_floor[c] = m - top_db;
}

hipError_t launch_clip_floor(const float *frame_max, long n_clips, int frames_per_clip, float top_db, float *clip_floor,
                             hipStream_t stream)
{
    if (n_clips <= 0) return hipSuccess;
    hipLaunchKernelGGL(clip_floor_kernel, dim3((unsigned)((n_clips + 255) / 256)), dim3(256), 0, stream, frame_max, n_clips,
                       frames_per_clip, top_db, clip_floor);
    return hipGetLastError();
}

int mfcc512_lds_bytes_per_block(bool tile) { return (int)lds_bytes<4, 10>(tile); }

template <int S, int L, int G, int TILE>
static int occupancy_one(bool full)
{
    int n = 0;
    const size_t lds = lds_bytes<S, L>(TILE);
    hipError_t e = full ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, mfcc512_wave_kernel<S, L, G, 512, 0, TILE, false>, 256, lds)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, mfcc512_wave_kernel<S, L, G, 0, 0, TILE, false>, 256, lds);
    return e == hipSuccess && n > 0 ? n : 4;
}

// resident 256-thread blocks per CU for the instantiation a plan will launch
int mfcc512_blocks_per_cu(int dct_split, int dct_len, int gather, bool full, bool tile)
{
#define DSP_OCC(S, L, G) \
    if (dct_split == S && dct_len == L && gather == G) return tile ? occupancy_one<S, L, G, 1>(full) : occupancy_one<S, L, G, 0>(full);
    DSP_FOR_SHAPES(DSP_OCC)
#undef DSP_OCC
    return 4;
}

}  // namespace dsp
