/*
 * ref_stop_shim.c -- accessors compiled TOGETHER WITH the reference's own
 * 2fa/audio/word/c/{mfcc.c, stop_detector.c, audio_classifier_inference.c} (from
 * /root/reference, never copied) into oracle/_ref/libref_stop.so.  It only exposes the
 * trained parameters of the reference's model_params.h (static const there) so the golden
 * generator can hand the same numbers to the oracle and to the HIP path.
 * TEST INFRASTRUCTURE ONLY; exists only where /root/reference does.
 */
#include "model_params.h" /* resolved via -I$(REF)/2fa/audio/word/c */

int ref_stop_input_size(void) { return INPUT_SIZE; }
void ref_stop_units(int *u) { u[0] = DENSE1_UNITS; u[1] = DENSE2_UNITS; u[2] = DENSE3_UNITS; u[3] = DENSE4_UNITS; }
const float *ref_stop_scaler_mean(void) { return SCALER_MEAN; }
const float *ref_stop_scaler_scale(void) { return SCALER_SCALE; }
const float *ref_stop_kernel(int layer)
{
    return layer == 0 ? DENSE1_KERNEL : layer == 1 ? DENSE2_KERNEL : layer == 2 ? DENSE3_KERNEL : DENSE4_KERNEL;
}
const float *ref_stop_bias(int layer)
{
    return layer == 0 ? DENSE1_BIAS : layer == 1 ? DENSE2_BIAS : layer == 2 ? DENSE3_BIAS : DENSE4_BIAS;
}
