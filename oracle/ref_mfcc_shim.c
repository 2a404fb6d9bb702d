/*
 * ref_mfcc_shim.c -- accessors compiled TOGETHER WITH the reference's own
 * 2fa/audio/word/c/mfcc.c (from /root/reference, never copied) into
 * oracle/_ref/libref_mfcc.so.  It only exposes the constant tables of the
 * reference's mfcc_params.h so tests can compare the oracle's analytic tables
 * with them.  TEST INFRASTRUCTURE ONLY; exists only where /root/reference does.
 */
#include "mfcc_params.h" /* resolved via -I$(REF)/2fa/audio/word/c */

const float *ref_hann_window(int *n) { *n = MFCC_FRAME_LENGTH; return HANN_WINDOW; }
const float *ref_mel_filter(int *rows, int *cols)
{
    *rows = MFCC_N_MELS; *cols = MFCC_N_FREQ_BINS; return MEL_FILTER;
}
const float *ref_dct_matrix(int *rows, int *cols)
{
    *rows = MFCC_N_MFCC; *cols = MFCC_N_MELS; return DCT_MATRIX;
}
