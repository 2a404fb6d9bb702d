// mfcc_kernels.hpp -- launch interface of mfcc_kernels.hip.
#pragma once

#include <hip/hip_runtime_api.h>

#include "clip_span.hpp"
#include "consumer_kernels.hpp"
#include "svm_kernels.hpp"
#include "tables.hpp"

namespace dsp {

// BASELINE config 5 in ONE kernel (clip mode, one wavefront per clip): the MFCC matrix is never written; the
// tile epilogue pools mean | std per coefficient (cepstrum/scrubjay_infer.c:36-66, float64 sums in frame order)
// and the clip ends with Scaler -> RBF-SVM -> Platt (scrubjay_svm.onnx, scrubjay_infer.c:105-141).
struct PoolSvmArgs {
    SvmModelDev svm;       // n_features = 2 * n_mfcc
    int *labels;           // [n_clips]
    float *decision;       // [n_clips] or nullptr
    float *prob1;          // [n_clips] or nullptr
    float *feat;           // [n_clips][2 * n_mfcc] or nullptr
};

// classify_signal in ONE kernel (2fa/audio/word/c/stop_detector.c:12-55): the tile epilogue standardises its
// 16 x 13 coefficients and adds their layer-1 products (audio_classifier_inference.c:18-36, 44-47) to float64 partial sums,
// the clip ends with layers 2-4 and the sigmoid -- the MFCC matrix never reaches HBM.
struct StopNetArgs {
    StopModelDev m;        // units[0] <= kStopFusedUnits, n_coef = n_mfcc <= 16
    float *prob;           // [n_clips]
};

struct Mfcc512Args {
    const void *in;                // HBM: frames or clips; float32, or int16 PCM (in_kind)
    int in_kind;                   // 0 float32 | 1 int16 mono | 2 int16 stereo ch 0 | 3 int16 stereo average
    float *out;                    // HBM: [n_frames][n_mfcc]
    const LaneTables512 *tables;   // HBM: per-lane constants
    long n_frames;                 // total frames over all clips
    long clip_stride;              // floats between clip starts (clip mode)
    int frames_per_clip;           // 0: independent frames back to back
    int hop;                       // clip mode: floats between frame starts
    int frame_len;                 // <= 512
    int chunk;                     // consecutive frames one wave takes at a time
    int n_mels, n_mfcc;
    float amin, top_db;
    // log_mode 1 (librosa power_to_db, ref = 1, top_db over the whole clip):
    //   frame_max != nullptr : pass 1, only write 10 log10(max mel energy) per frame, no MFCC
    //   clip_floor != nullptr: pass 2, clip log-mel at clip_floor[clip]
    //   both nullptr         : every frame is its own clip (independent frames), one pass
    // log_mode 2 (DSP_LOG_LOG10_FLOOR, n_fft 2048): plain log10 with aubio's 2e-42 floor
    int log_mode;
    // n_fft 2048 only (the aubio-semantics front end of cepstrum/scrubjay_infer.c): spectrum 1 = magnitude into the filterbank;
    // stream_framing: frame t of a clip covers samples [(t + 1) hop - frame_len, (t + 1) hop), zeros outside [0, samples_per_clip)
    int spectrum = 0;
    int stream_framing = 0;
    int samples_per_clip = 0;
    float *frame_max;
    const float *clip_floor;
    // the fused clip kernels (POOL; one wavefront walks one clip): the clip count, and for a ragged batch the clips' spans
    // (spans == nullptr: clip c starts at c * clip_stride and has frames_per_clip frames of samples_per_clip samples)
    long n_clips = 0;
    const ClipSpan *spans = nullptr;
    PoolSvmArgs pool;              // only read by the POOL instantiations (launch_mfcc512_pool)
    StopNetArgs stop{};            // only read by the stop-net instantiations (launch_mfcc512_stop)
};

// clip_floor[c] = max_t frame_max[c][t] - top_db
hipError_t launch_clip_floor(const float *frame_max, long n_clips, int frames_per_clip, float top_db, float *clip_floor,
                             hipStream_t stream);

// tile: 16-frame log + MFMA-DCT epilogue (per-frame log mode, chunk % 8 == 0); false: per-frame epilogue
hipError_t launch_mfcc512(const Mfcc512Args &args, int dct_split, int dct_len, int gather, int blocks,
                          hipStream_t stream, bool tile);
// fused clip -> label path: args.chunk must equal args.frames_per_clip, args.out is not written
hipError_t launch_mfcc512_pool(const Mfcc512Args &args, int dct_split, int dct_len, int gather, int blocks, hipStream_t stream);
// fused clip -> stop-word probability (reference shape only: 13 coefficients, 40 mel): args.chunk == args.frames_per_clip, args.stop set
int mfcc512_stop_grid(int blocks, const StopModelDev &m, int in_kind, int gather, int frame_len);      // the grid launch_mfcc512_stop starts for a caller's count
hipError_t launch_mfcc512_stop(const Mfcc512Args &args, int dct_split, int dct_len, int gather, int blocks, hipStream_t stream);
hipError_t launch_mfcc512_row(const Mfcc512Args &args, const RowTables512 *row_tables, int dct_split, int dct_len,
                              int gather, int blocks, hipStream_t stream);
int mfcc512_row_blocks_per_cu(int dct_split, int dct_len, int gather, bool full);
hipError_t launch_mfcc1024(const Mfcc512Args &args, const GenTables1024 *tables, int blocks, hipStream_t stream);
int mfcc1024_blocks_per_cu(bool full);
// register-resident wave-per-frame form (mfcc1024_wave_kernel.hip): tables->n_chunk_slots <= 3, chunk % 8 == 0
struct PrefilterScan;   // tables.hpp
// scan != nullptr: independent 1024-sample frames are band-pass filtered (PrefilterScan, tables.hpp) inside the kernel
// scan_steps (host copy of PrefilterScan::c_steps, with scan): picks the instantiation whose compile-time step counts cover them
hipError_t launch_mfcc1024_wave(const Mfcc512Args &args, const GenTables1024 *tables, int blocks, hipStream_t stream,
                                const PrefilterScan *scan = nullptr, const int *scan_steps = nullptr);
int mfcc1024_wave_blocks_per_cu(bool full, bool prefilter = false);
int mfcc512_lds_bytes_per_block(bool tile);
// n_fft = 2048 (mfcc2048_kernel.hip); pool: the fused clip -> label form (args.chunk == args.frames_per_clip, args.pool set)
struct GenTables2048;
struct PairExtra512;
// two frames per wavefront step, reference shape, independent 512-sample frames (mfcc512_pair_kernel.hip; DSP_KERNEL_PAIR)
hipError_t launch_mfcc512_pair(const Mfcc512Args &args, const PairExtra512 *extra, int blocks, hipStream_t stream);
int mfcc512_pair_blocks_per_cu();
hipError_t launch_mfcc2048(const Mfcc512Args &args, const GenTables2048 *tables, int blocks, hipStream_t stream, bool pool);
int mfcc2048_blocks_per_cu(int n_mels, bool pool, bool aub = false);      // aub: the aubio-semantics kernels (spectrum / log10 / stream framing) and their LDS
int mfcc512_blocks_per_cu(int dct_split, int dct_len, int gather, bool full, bool tile);

}  // namespace dsp
