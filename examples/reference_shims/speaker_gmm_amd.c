/*
 * speaker_gmm_amd.c -- what a maintainer adds next to the reference's 2fa/audio/pico-audio/src/speaker_gmm.h to run
 * the speaker verification on the MI355X: it replaces speaker_gmm.c in a host build; the trained
 * tables stay in the reference's own gmm_params.inc.
 *
 *   gcc -O2 -I<reference>/2fa/audio/pico-audio/src -I<dsp_amd>/include caller.c speaker_gmm_amd.c \
 *       -L<dsp_amd>/dsp_amd -ldsp_amd -L/opt/rocm/lib -lamdhip64
 *
 * Declarations kept: speaker_gmm.h:36-38.  mfcc_feats is frame-major [num_frames][13], as
 * speaker_gmm.c:131 indexes it.
 */
#include "speaker_gmm.h"

#include "gmm_params.inc" /* K, D, target_* / ubm_* tables */

#include <dsp_amd.h>
#include <hip/hip_runtime_api.h>
#include <stdio.h>

static dsp_speaker_model *model(void)
{
    static dsp_speaker_model *m;
    if (!m) {
        dsp_gmm_params t = {K, D, &target_means[0][0], &target_inv_covs[0][0], target_log_consts};
        dsp_gmm_params u = {K, D, &ubm_means[0][0], &ubm_inv_covs[0][0], ubm_log_consts};
        if (dsp_speaker_model_create(&t, &u, 0, &m) != DSP_OK) fprintf(stderr, "speaker model: %s\n", dsp_last_error());
    }
    return m;
}

static int run(float *mfcc_feats, int num_frames, int64_t *llr_mean, int *label)
{
    dsp_speaker_model *m = model();
    float *d_mfcc = NULL;
    int64_t *d_llr = NULL;
    int *d_label = NULL;
    int ok = 0;
    const size_t bytes = (size_t)num_frames * D * sizeof(float);
    if (m && num_frames > 0 && hipMalloc((void **)&d_mfcc, bytes) == hipSuccess && hipMalloc((void **)&d_llr, sizeof(int64_t)) == hipSuccess &&
        hipMalloc((void **)&d_label, sizeof(int)) == hipSuccess && hipMemcpy(d_mfcc, mfcc_feats, bytes, hipMemcpyHostToDevice) == hipSuccess &&
        dsp_speaker_llr_device(m, d_mfcc, 1, num_frames, d_llr, d_label, NULL, NULL, NULL) == DSP_OK &&
        hipMemcpy(llr_mean, d_llr, sizeof(int64_t), hipMemcpyDeviceToHost) == hipSuccess &&
        hipMemcpy(label, d_label, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess)
        ok = 1;
    if (d_mfcc) hipFree(d_mfcc);
    if (d_llr) hipFree(d_llr);
    if (d_label) hipFree(d_label);
    return ok;
}

int64_t mfcc_target_speaker_llr_mean(float *mfcc_feats, int num_frames)
{
    int64_t mean = 0;
    int label = 0;
    if (!run(mfcc_feats, num_frames, &mean, &label)) fprintf(stderr, "mfcc_target_speaker_llr_mean: %s\n", dsp_last_error());
    return mean;
}

int classify_speaker(float *mfcc_feats, int num_frames)
{
    int64_t mean = 0;
    int label = 0;
    if (!run(mfcc_feats, num_frames, &mean, &label)) fprintf(stderr, "classify_speaker: %s\n", dsp_last_error());
    return label;
}
