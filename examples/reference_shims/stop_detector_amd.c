/*
 * stop_detector_amd.c -- what a maintainer adds to the reference's 2fa/audio/word/c/ to run classify_signal()
 * on the MI355X: it replaces stop_detector.c + audio_classifier_inference.c + mfcc.c in the
 * host build.  The trained parameters stay in the reference's own model_params.h and are handed to the library once.
 *
 *   gcc -O2 -I<reference>/2fa/audio/word/c -I<dsp_amd>/include main_test.c stop_detector_amd.c \
 *       -L<dsp_amd>/dsp_amd -ldsp_amd -lm
 *
 * Declarations kept: stop_detector.h:10,13-15.
 */
#include "stop_detector.h"

#include "mfcc_params.h"  /* MFCC_N_MFCC */
#include "model_params.h" /* INPUT_SIZE, DENSEn_UNITS, SCALER_*, DENSEn_KERNEL / _BIAS */

#include <dsp_amd.h>
#include <stdio.h>

static dsp_stop_model *model(void)
{
    static dsp_stop_model *m;
    if (!m) {
        dsp_stop_model_params p;
        p.n_coef = MFCC_N_MFCC;
        p.max_frames = INPUT_SIZE / MFCC_N_MFCC; /* MAX_FRAMES of stop_detector.c:9 */
        p.units[0] = DENSE1_UNITS; p.units[1] = DENSE2_UNITS; p.units[2] = DENSE3_UNITS; p.units[3] = DENSE4_UNITS;
        p.scaler_mean = SCALER_MEAN;
        p.scaler_scale = SCALER_SCALE;
        p.kernel[0] = DENSE1_KERNEL; p.kernel[1] = DENSE2_KERNEL; p.kernel[2] = DENSE3_KERNEL; p.kernel[3] = DENSE4_KERNEL;
        p.bias[0] = DENSE1_BIAS; p.bias[1] = DENSE2_BIAS; p.bias[2] = DENSE3_BIAS; p.bias[3] = DENSE4_BIAS;
        if (dsp_stop_model_create(&p, 0, &m) != DSP_OK) fprintf(stderr, "classify_signal: %s\n", dsp_last_error());
    }
    return m;
}

float classify_signal(const float *signal, int num_samples)
{
    dsp_stop_model *m = model();
    return m ? dsp_classify_signal(m, signal, num_samples) : 0.0f;
}

int classify_signal_binary(const float *signal, int num_samples, float threshold)
{
    return classify_signal(signal, num_samples) > threshold ? 1 : 0;
}
