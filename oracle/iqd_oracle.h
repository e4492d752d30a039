/* oracle/iqd_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the RtlSdrDiags per-sample DSP hot path
 * (IqDataProcessor::acceptIqData -> AM / FM / WBFM / SSB chains -> 8 kS/s PCM).
 * It exists to check the HIP engine; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product path never does.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_golden.py checks it bit-for-bit
 * against the .npz fixtures under tests/golden, which were produced by the unmodified reference
 * sources compiled here (oracle/_ref, recipe in oracle/Makefile) by
 * tests/golden/make_golden.py; where oracle/_ref is present the tests also
 * compare the two live on fresh random inputs.
 */
#ifndef IQD_ORACLE_H
#define IQD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* demodulatorType, hdr_diags/IqDataProcessor.h:20 */
enum { IQO_NONE = 0, IQO_AM = 1, IQO_FM = 2, IQO_WBFM = 3, IQO_LSB = 4, IQO_USB = 5 };

typedef struct iqo_chain iqo_chain;

iqo_chain *iqo_create(void);
void iqo_destroy(iqo_chain *c);
void iqo_reset(iqo_chain *c);                       /* all four demodulators */
void iqo_reset_demod(iqo_chain *c, int which);      /* one of them: 1 AM 2 FM 3 WBFM 4 SSB */
void iqo_set_mode(iqo_chain *c, int mode);
void iqo_set_gain(iqo_chain *c, int which, float gain); /* 1 AM 2 FM 3 WBFM 4 SSB */
void iqo_set_squelch(iqo_chain *c, int32_t threshold);
void iqo_set_rx_gain_db(iqo_chain *c, uint32_t gain_db);
/* +1: +Fs/4 (what acceptIqData applies), -1: -Fs/4, 0: none (extension knob) */
void iqo_set_rotation(iqo_chain *c, int rotation);

/* AutomaticGainControl (src_diags/AutomaticGainControl.cc): one per chain, constructed like Radio.cc:184
 * (operating point -12 dBFS), disabled.  Setters return the reference's success flag. */
enum { IQO_AGC_LOWPASS = 0, IQO_AGC_HARRIS = 1 };
typedef struct iqo_agc {
    uint32_t type;
    uint32_t blanking_counter, blanking_limit;
    int gain_was_adjusted;
    int32_t deadband_db;
    int enabled;
    int32_t operating_point_dbfs;
    float alpha;
    uint32_t if_gain_db;
    float filtered_if_gain_db;
    uint32_t signal_magnitude;
    int32_t normalized_level_dbfs;
} iqo_agc;
iqo_agc *iqo_agc_of(iqo_chain *c);
void iqo_agc_feed(iqo_chain *c, uint32_t magnitude);   /* one magnitude callback */
uint32_t iqo_get_rx_gain_db(const iqo_chain *c);
int iqo_agc_set_type(iqo_chain *c, uint32_t type);
int iqo_agc_set_deadband(iqo_chain *c, uint32_t deadband_db);
int iqo_agc_set_blanking_limit(iqo_chain *c, uint32_t limit);
void iqo_agc_set_operating_point(iqo_chain *c, int32_t dbfs);
int iqo_agc_set_filter_coefficient(iqo_chain *c, float coefficient);
int iqo_agc_enable(iqo_chain *c, int on);
/* FrequencyScanner (src_diags/FrequencyScanner.cc): one per chain, idle, 162.55 MHz.  tuned_hz / tune_count record
 * the Radio::setReceiveFrequency commands it issues. */
typedef struct iqo_scanner {
    uint64_t start_hz, end_hz, increment_hz, current_hz;
    int new_configuration, scanning;
    uint64_t tuned_hz;
    uint32_t tune_count;
} iqo_scanner;
iqo_scanner *iqo_scanner_of(iqo_chain *c);
int iqo_scanner_set_parameters(iqo_chain *c, uint64_t start_hz, uint64_t end_hz, uint64_t increment_hz);
int iqo_scanner_start(iqo_chain *c);
int iqo_scanner_stop(iqo_chain *c);
void iqo_scanner_feed(iqo_chain *c, int signal_present);   /* one signal-state callback */
/* One AutomaticGainControl::run(magnitude) with the radio's IF gain `gain`; returns the gain afterwards. */
uint32_t iqo_agc_run(iqo_agc *a, uint32_t magnitude, uint32_t gain);

/* One acceptIqData() call.  bytes must be a multiple of 8 and <= 32768.
 * Returns the number of PCM samples produced (0 when squelched / mode None). */
long iqo_accept(iqo_chain *c, const uint8_t *iq, size_t bytes,
                int16_t *pcm, size_t pcm_capacity,
                uint32_t *magnitude, uint8_t *allowed);

/* Feeds `total` bytes in calls of `block_bytes`; per-block magnitude/allowed. */
long iqo_accept_stream(iqo_chain *c, const uint8_t *iq, size_t total, size_t block_bytes,
                       int16_t *pcm, size_t pcm_capacity,
                       uint32_t *magnitude, uint8_t *allowed);

/* Demodulator-level entry on signed, already rotated samples (no front end,
 * no squelch) — the shape of demodulatorResearch/demodulators/demod.cc. */
long iqo_demod_accept(iqo_chain *c, int mode, const int8_t *iq, size_t bytes,
                      int16_t *pcm, size_t pcm_capacity);

/* ---- primitives, exported for known-answer tests ---- */
void iqo_quantize_taps(const float *h, int length, int16_t *hq);
long iqo_decimate_q15(const float *h, int length, int factor,
                      const int16_t *in, size_t n, int16_t *out);   /* zero state */
void iqo_fir_q15(const float *h, int length, const int16_t *in, size_t n, int16_t *out);
void iqo_fir_f32(const float *h, int length, const float *in, size_t n, float *out);
void iqo_iir_f32(const float *b, int nb, const float *a, int na,
                 const float *in, size_t n, float *out);
/* Float Decimator / Interpolator and Interpolator_int16 (Filters/), whole streams from the zero state */
long iqo_decimate_f32(const float *h, int length, int factor, const float *in, size_t n, float *out);
void iqo_interpolate_f32(const float *h, int length, int factor, const float *in, size_t n, float *out);
void iqo_interpolate_q15(const float *h, int length, int factor, const int16_t *in, size_t n, int16_t *out);
void iqo_rotate(int8_t *buf, size_t bytes, int rotation);
void iqo_atan2_lut(float *lut /* [256][256], lut[y][x] */);
void iqo_fm_theta_lut(float *lut, int half_range /* lut[(q+R)*(2R+1)+(i+R)] */);
void iqo_db_table(int32_t *table /* [257] */);
int32_t iqo_dbfs(uint32_t magnitude);
int16_t iqo_cast_i16(float f);
uint32_t iqo_block_magnitude(const int8_t *s, size_t bytes);

/* Q15 tap tables as the reference constructors quantise them (debug / tests).
 * which: 0 wbfm_pre 1 wbfm_d1 2 wbfm_d2 3 audio40 4 fm_tuner 5 fm_post
 *        6 am_s1 7 am_s2 8 am_s3 9 ssb_delay 10 ssb_hilbert.  Returns length. */
int iqo_get_taps_q15(int which, int16_t *hq);
int iqo_get_taps_f32(int which, float *h);

/* WBFM stage taps for debugging the HIP kernels: runs the WBFM chain on signed,
 * rotated samples from zero state and returns the intermediate arrays. */
void iqo_wbfm_stages(const int8_t *rot, size_t n_samples, float gain,
                     int8_t *ip, int8_t *qp, float *theta, float *dtheta,
                     float *deemph, int16_t *w);

#ifdef __cplusplus
}
#endif
#endif
