// mfcc_kernels.hpp -- launch interface of mfcc_kernels.hip.
#pragma once

#include <hip/hip_runtime_api.h>

#include "tables.hpp"

namespace dsp {

struct Mfcc512Args {
    const float *in;               // HBM: frames or clips
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
};

hipError_t launch_mfcc512(const Mfcc512Args &args, int dct_split, int dct_len, int gather, int blocks,
                          hipStream_t stream);
hipError_t launch_mfcc512_row(const Mfcc512Args &args, const RowTables512 *row_tables, int dct_split, int dct_len,
                              int gather, int blocks, hipStream_t stream);
int mfcc512_row_blocks_per_cu(int dct_split, int dct_len, int gather, bool full);
hipError_t launch_mfcc1024(const Mfcc512Args &args, const GenTables1024 *tables, int blocks, hipStream_t stream);
int mfcc1024_blocks_per_cu(bool full);
int mfcc512_lds_bytes_per_block();
int mfcc512_frames_per_item();   // frames a wave processes together; chunk must be a multiple
int mfcc512_blocks_per_cu(int dct_split, int dct_len, int gather, bool full);

}  // namespace dsp
