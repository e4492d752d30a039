/* oracle/iqd_oracle.c — TEST INFRASTRUCTURE ONLY (see iqd_oracle.h).
 *
 * CPU restatement of the RtlSdrDiags hot path, written block-wise (whole-array
 * stages with explicit history) instead of the reference's per-sample ring
 * buffers.  Every function cites the reference file:line it follows; paths are
 * relative to /root/reference/radioDiags.  Build: gcc -O3 -ffp-contract=off.
 *
 * Platform rules reproduced here (x86-64 / gcc, the reference's only target):
 *   (int16_t)float  = cvttss2si to int32 (0x80000000 when out of range), then
 *                     keep the low 16 bits                    -> iqo_cast_i16
 *   int8/int16 narrowing wraps (two's complement)
 *   float expressions are evaluated op by op in binary32, no FMA contraction
 */
#include "iqd_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define MAX_BLOCK_SAMPLES 16384 /* hdr_diags/SignalDetector.h:49 caps a call at 32768 B */
#define QMAX 40

/* ------------------------------------------------------------------------ */
/* Tap tables.  Every FIR in the path is linear phase, so only the first half
 * of each design is listed and expand_taps() mirrors it (odd = antisymmetric).
 * Values: WbFmDemodulator.cc:17-110, FmDemodulator.cc:14-108,
 * AmDemodulator.cc:14-62, SsbDemodulator.cc:14-100.                          */
/* ------------------------------------------------------------------------ */
static const float H_WBFM_PRE[8] = {-0.0157211f, -0.0325959f, 0.0092996f, 0.0621217f,
                                    -0.0148595f, -0.0989456f, 0.1182989f, 0.4862333f};
static const float H_WBFM_D1[4] = {0.0243699f, 0.0769537f, 0.1463572f, 0.1967096f};
static const float H_POST12[6] = {0.0022977f, 0.0237042f, 0.0605386f,
                                  0.1127073f, 0.1645167f, 0.1971107f};
static const float H_AUDIO40[20] = {
    0.0015969f, -0.0111080f, -0.0270501f, -0.0265610f, -0.0023190f, 0.0180618f, 0.0065495f,
    -0.0183409f, -0.0133345f, 0.0184489f, 0.0230891f, -0.0161248f, -0.0363745f, 0.0091343f,
    0.0550219f, 0.0070312f, -0.0862280f, -0.0497761f, 0.1793543f, 0.4145808f};
static const float H_FM_TUNER[16] = {0.0041331f, 0.0054174f, 0.0076016f, 0.0115481f,
                                     0.0151685f, 0.0203192f, 0.0251608f, 0.0311322f,
                                     0.0366372f, 0.0427168f, 0.0480527f, 0.0533425f,
                                     0.0575831f, 0.0611914f, 0.0635413f, 0.0648239f};
static const float H_AM_S1[4] = {0.0242683f, 0.0766338f, 0.1457589f, 0.1959036f};
static const float H_AM_S2[6] = {0.0057496f, 0.0263853f, 0.0605301f,
                                 0.1074406f, 0.1523486f, 0.1804951f};
static const float H_AM_S3[8] = {0.0116487f, 0.0152694f, -0.0109804f, -0.0611915f,
                                 -0.0736143f, 0.0187617f, 0.1988190f, 0.3481364f};
/* 31-tap Hamming-windowed Hilbert transformer: even indices below, odd indices
 * and the centre are zero, second half is the negated mirror. */
static const float H_SSB_HILBERT_EVEN[8] = {-0.0033953f, -0.0058652f, -0.0134385f, -0.0281423f,
                                            -0.0534836f, -0.0980394f, -0.1935638f, -0.6302204f};

/* De-emphasis (75 us) and DC-removal IIR sections:
 * WbFmDemodulator.cc:114-123, AmDemodulator.cc:67-68, SsbDemodulator.cc:105-106 */
static const float B_DEEMPH[2] = {0.0253863f, 0.0253863f};
static const float A_DEEMPH[1] = {-0.9492274f};
static const float B_DCBLOCK[2] = {1.0f, -1.0f};
static const float A_DCBLOCK[1] = {-0.95f};

enum { T_WBFM_PRE, T_WBFM_D1, T_WBFM_D2, T_AUDIO40, T_FM_TUNER, T_FM_POST,
       T_AM_S1, T_AM_S2, T_AM_S3, T_SSB_DELAY, T_SSB_HILBERT, T_COUNT };

static int expand_taps(int which, float *h)
{
    const float *half = 0;
    int n = 0, i;
    switch (which) {
    case T_WBFM_PRE: half = H_WBFM_PRE; n = 8; break;
    case T_WBFM_D1: half = H_WBFM_D1; n = 4; break;
    case T_WBFM_D2: case T_FM_POST: half = H_POST12; n = 6; break;
    case T_AUDIO40: half = H_AUDIO40; n = 20; break;
    case T_FM_TUNER: half = H_FM_TUNER; n = 16; break;
    case T_AM_S1: half = H_AM_S1; n = 4; break;
    case T_AM_S2: half = H_AM_S2; n = 6; break;
    case T_AM_S3: half = H_AM_S3; n = 8; break;
    case T_SSB_DELAY: /* SsbDemodulator.cc:65: fifteen zeros then 1 */
        for (i = 0; i < 15; i++) h[i] = 0.0f;
        h[15] = 1.0f;
        return 16;
    case T_SSB_HILBERT: /* SsbDemodulator.cc:68-100 */
        for (i = 0; i < 31; i++) h[i] = 0.0f;
        for (i = 0; i < 8; i++) {
            h[2 * i] = H_SSB_HILBERT_EVEN[i];
            h[30 - 2 * i] = -H_SSB_HILBERT_EVEN[i];
        }
        return 31;
    default: return 0;
    }
    for (i = 0; i < n; i++) {
        h[i] = half[i];
        h[2 * n - 1 - i] = half[i];
    }
    return 2 * n;
}

/* ------------------------------------------------------------------------ */
/* Scalar rules                                                               */
/* ------------------------------------------------------------------------ */

/* (int16_t)f as x86-64 gcc executes it (AmDemodulator.cc:465,
 * SsbDemodulator.cc:592, FmDemodulator.cc:540, WbFmDemodulator.cc:536). */
int16_t iqo_cast_i16(float f)
{
    int32_t wide;
    if (f >= -2147483648.0f && f < 2147483648.0f)
        wide = (int32_t)f; /* truncation toward zero */
    else
        wide = INT32_MIN; /* "integer indefinite", also for NaN */
    return (int16_t)(uint16_t)((uint32_t)wide & 0xffffu);
}

/* Decimator_int16.cc:55-63 / FirFilter_int16.cc:46-54:
 * hq = (int16_t) round(h * 32768) with the cast rule above (1.0 -> -32768). */
void iqo_quantize_taps(const float *h, int length, int16_t *hq)
{
    int i;
    for (i = 0; i < length; i++) {
        float scaled = h[i] * 32768;
        scaled = roundf(scaled);
        hq[i] = iqo_cast_i16(scaled);
    }
}

/* ------------------------------------------------------------------------ */
/* Q15 FIR / decimator with explicit history                                  */
/* ------------------------------------------------------------------------ */
typedef struct {
    int length;           /* L */
    int factor;           /* M (1 = plain FIR) */
    int phase;            /* inputs seen so far, mod M (Decimator_int16.cc:320-328) */
    int16_t hq[QMAX];
    int16_t hist[QMAX];   /* the last L-1 inputs, oldest first */
} q15_filter;

static void q15_init(q15_filter *f, int which, int factor)
{
    float h[QMAX];
    memset(f, 0, sizeof(*f));
    f->length = expand_taps(which, h);
    f->factor = factor;
    iqo_quantize_taps(h, f->length, f->hq);
}

static void q15_init_taps(q15_filter *f, const float *h, int length, int factor)
{
    memset(f, 0, sizeof(*f));
    f->length = length;
    f->factor = factor;
    iqo_quantize_taps(h, length, f->hq);
}

static void q15_reset(q15_filter *f)
{
    memset(f->hist, 0, sizeof(f->hist));
    f->phase = 0;
}

/* One Q15 dot product, Decimator_int16.cc:176-238 / FirFilter_int16.cc:151-213:
 * accumulator starts at 1<<14, is clamped after every MAC, result is >>15. */
static inline int16_t q15_dot(const int16_t *hq, int length, const int16_t *newest)
{
    int32_t acc = 1 << 14;
    int k;
    for (k = 0; k < length; k++) {
        acc = acc + (int32_t)hq[k] * (int32_t)newest[-k];
        if (acc > 0x3fffffff)
            acc = 0x3fffffff;
        else if (acc < -0x40000000)
            acc = -0x40000000;
    }
    return (int16_t)(acc >> 15);
}

/* Runs n inputs through the filter; an output appears for every input whose
 * running index is M-1 (mod M) (Decimator_int16.cc:310-351).  `work` must hold
 * (L-1)+n samples.  Returns the number of outputs. */
static size_t q15_run(q15_filter *f, const int16_t *in, size_t n, int16_t *out, int16_t *work)
{
    const int L = f->length, M = f->factor;
    size_t i, m = 0;
    memcpy(work, f->hist, (size_t)(L - 1) * sizeof(int16_t));
    memcpy(work + (L - 1), in, n * sizeof(int16_t));
    for (i = (size_t)(M - 1 - f->phase); i < n; i += (size_t)M)
        out[m++] = q15_dot(f->hq, L, work + (L - 1) + i);
    f->phase = (int)((f->phase + n) % (size_t)M);
    memmove(f->hist, work + n, (size_t)(L - 1) * sizeof(int16_t));
    return m;
}

/* ------------------------------------------------------------------------ */
/* Float FIR / IIR                                                            */
/* ------------------------------------------------------------------------ */
typedef struct {
    int length;
    float h[8];
    float x[8]; /* x[0] = previous input, x[1] = the one before, ... */
} f32_fir;

static void f32_fir_init(f32_fir *f, const float *h, int length)
{
    memset(f, 0, sizeof(*f));
    f->length = length;
    memcpy(f->h, h, (size_t)length * sizeof(float));
}

/* FirFilter.cc:144-185: y = 0; y = y + h[k]*x[n-k] for k = 0..L-1, in order. */
static inline float f32_fir_step(f32_fir *f, float x)
{
    float y = 0;
    int k;
    y = y + (f->h[0] * x);
    for (k = 1; k < f->length; k++)
        y = y + (f->h[k] * f->x[k - 1]);
    for (k = f->length - 2; k > 0; k--)
        f->x[k] = f->x[k - 1];
    if (f->length > 1)
        f->x[0] = x;
    return y;
}

typedef struct {
    f32_fir num;
    int na;
    float a[4];
    float y[4]; /* y[0] = previous output */
} f32_iir;

static void f32_iir_init(f32_iir *f, const float *b, int nb, const float *a, int na)
{
    memset(f, 0, sizeof(*f));
    f32_fir_init(&f->num, b, nb);
    f->na = na;
    memcpy(f->a, a, (size_t)na * sizeof(float));
}

/* IirFilter.cc:161-176,199-229: y = FIR_b(x); y -= (0 + a[0]*y[n-1] + ...). */
static inline float f32_iir_step(f32_iir *f, float x)
{
    float y = f32_fir_step(&f->num, x);
    float r = 0;
    int k;
    for (k = 0; k < f->na; k++)
        r = r + (f->a[k] * f->y[k]);
    y -= r;
    for (k = f->na - 1; k > 0; k--)
        f->y[k] = f->y[k - 1];
    f->y[0] = y;
    return y;
}

static void f32_iir_reset(f32_iir *f)
{
    memset(f->num.x, 0, sizeof(f->num.x));
    memset(f->y, 0, sizeof(f->y));
}

/* ------------------------------------------------------------------------ */
/* Tables                                                                     */
/* ------------------------------------------------------------------------ */
static float g_atan2_lut[256][256];
static int32_t g_db_table[257];
static int g_tables_ready = 0;

/* WbFmDemodulator.cc:159-170: lut[y][x] = (float)atan2(y-128, x-128). */
void iqo_atan2_lut(float *lut)
{
    int x, y;
    for (x = 0; x < 256; x++)
        for (y = 0; y < 256; y++)
            lut[y * 256 + x] = (float)atan2((double)y - 128, (double)x - 128);
}

/* FmDemodulator.cc:476 evaluated for every (q, i) the tuner decimator can emit
 * from int8 input: theta = (float)atan2((double)q, (double)i). */
void iqo_fm_theta_lut(float *lut, int R)
{
    int q, i, W = 2 * R + 1;
    for (q = -R; q <= R; q++)
        for (i = -R; i <= R; i++)
            lut[(q + R) * W + (i + R)] = (float)atan2((double)q, (double)i);
}

/* DbfsCalculator.cc:36-68 with wordLengthInBits = 7 (SignalDetector.cc:34).
 * The reference is C++: log10((float)i) resolves to the float overload. */
void iqo_db_table(int32_t *table)
{
    uint32_t i;
    for (i = 1; i <= 256; i++) {
        float level = 20 * log10f((float)i);
        table[i] = (int32_t)level;
    }
    table[0] = table[1];
}

static void tables_init(void)
{
    if (g_tables_ready) return;
    iqo_atan2_lut(&g_atan2_lut[0][0]);
    iqo_db_table(g_db_table);
    g_tables_ready = 1;
}

/* DbfsCalculator.cc:111-147 */
int32_t iqo_dbfs(uint32_t magnitude)
{
    const uint32_t full_scale = (1u << 7) - 1;
    const uint32_t full_scale_db = (uint32_t)(20 * log10((double)full_scale));
    int32_t decibels = 0, value;
    tables_init();
    if (magnitude > full_scale) magnitude = full_scale;
    while (magnitude > 256) {
        magnitude /= 2;
        decibels += 6;
    }
    value = g_db_table[magnitude];
    value += decibels;
    value -= (int32_t)full_scale_db;
    return value;
}

/* ------------------------------------------------------------------------ */
/* Front end                                                                  */
/* ------------------------------------------------------------------------ */

/* IqDataProcessor.cc:567-611 (+Fs/4) and :496-540 (-Fs/4); the phase restarts
 * at the start of every call; int8 negation wraps (-(-128) == -128). */
void iqo_rotate(int8_t *b, size_t bytes, int rotation)
{
    size_t i;
    if (rotation == 0) return;
    for (i = 0; i + 8 <= bytes; i += 8) {
        int8_t x, y;
        x = b[i + 2]; y = b[i + 3];
        if (rotation > 0) { b[i + 2] = (int8_t)-y; b[i + 3] = x; }
        else              { b[i + 2] = y; b[i + 3] = (int8_t)-x; }
        x = b[i + 4]; y = b[i + 5];
        b[i + 4] = (int8_t)-x; b[i + 5] = (int8_t)-y;
        x = b[i + 6]; y = b[i + 7];
        if (rotation > 0) { b[i + 6] = y; b[i + 7] = (int8_t)-x; }
        else              { b[i + 6] = (int8_t)-y; b[i + 7] = x; }
    }
}

/* SignalDetector.cc:227-259: mean over the call of max + min/2 of |I|,|Q|
 * (uint8 arithmetic, tie goes to the Q branch), integer division. */
uint32_t iqo_block_magnitude(const int8_t *s, size_t bytes)
{
    uint32_t sum = 0;
    size_t i, n = bytes / 2;
    for (i = 0; i < bytes; i += 2) {
        uint8_t a = (uint8_t)abs((int)s[i]);
        uint8_t b = (uint8_t)abs((int)s[i + 1]);
        uint8_t m = (a > b) ? (uint8_t)(a + (b >> 1)) : (uint8_t)(b + (a >> 1));
        sum += m;
    }
    return n ? sum / (uint32_t)n : 0;
}

/* ------------------------------------------------------------------------ */
/* Demodulators                                                               */
/* ------------------------------------------------------------------------ */
typedef struct {
    q15_filter pre_i, pre_q, d1, d2, d3;
    f32_iir deemph;
    float theta_prev;
    float gain;
} wbfm_t;

typedef struct {
    q15_filter tuner_i, tuner_q, post, audio;
    f32_fir diff;
    float gain;
} fm_t;

typedef struct {
    q15_filter s1i, s1q, s2i, s2q, s3i, s3q;
    f32_iir dc;
    float gain;
} am_t;

typedef struct {
    q15_filter s1i, s1q, s2i, s2q, s3i, s3q, delay, hilbert;
    f32_iir dc;
    float gain;
    int lsb;
} ssb_t;

struct iqo_chain {
    int mode;
    int rotation;
    int32_t threshold;
    uint32_t rx_gain_db;
    int tracking; /* SignalTracker state: 0 NoSignal, 1 Tracking */
    iqo_agc agc;
    iqo_scanner scanner;
    wbfm_t wbfm;
    fm_t fm;
    am_t am;
    ssb_t ssb;
    /* scratch */
    int8_t s8[2 * MAX_BLOCK_SAMPLES];
    int16_t ri[MAX_BLOCK_SAMPLES], rq[MAX_BLOCK_SAMPLES];
    int16_t a16[MAX_BLOCK_SAMPLES], b16[MAX_BLOCK_SAMPLES];
    int16_t c16[MAX_BLOCK_SAMPLES], d16[MAX_BLOCK_SAMPLES];
    int16_t work[MAX_BLOCK_SAMPLES + QMAX];
};

static void split_rails(const int8_t *iq, size_t n, int16_t *ri, int16_t *rq)
{
    size_t i;
    for (i = 0; i < n; i++) {
        ri[i] = (int16_t)iq[2 * i];
        rq[i] = (int16_t)iq[2 * i + 1];
    }
}

/* Delta-theta branch-cut handling, WbFmDemodulator.cc:472-480 and
 * FmDemodulator.cc:485-493: comparisons and the +-2*pi step are in double. */
static inline float wrap_delta(float d)
{
    while (d > M_PI) d -= (2 * M_PI);
    while (d < (-M_PI)) d += (2 * M_PI);
    return d;
}

static void wbfm_init(wbfm_t *w)
{
    q15_init(&w->pre_i, T_WBFM_PRE, 1);
    q15_init(&w->pre_q, T_WBFM_PRE, 1);
    q15_init(&w->d1, T_WBFM_D1, 4);
    q15_init(&w->d2, T_WBFM_D2, 4);
    q15_init(&w->d3, T_AUDIO40, 2);
    f32_iir_init(&w->deemph, B_DEEMPH, 2, A_DEEMPH, 1);
    w->theta_prev = 0;
    w->gain = 256000 / (2 * M_PI); /* WbFmDemodulator.cc:173 */
}

/* WbFmDemodulator.cc:304-320.  Note: the de-emphasis filter is NOT reset. */
static void wbfm_reset(wbfm_t *w)
{
    q15_reset(&w->pre_i); q15_reset(&w->pre_q);
    q15_reset(&w->d1); q15_reset(&w->d2); q15_reset(&w->d3);
    w->theta_prev = 0;
}

/* WbFmDemodulator.cc:383-411 (acceptIqData), :436-494 (demodulateSignal),
 * :515-562 (createPcmData). */
static size_t wbfm_run(iqo_chain *c, wbfm_t *w, const int8_t *iq, size_t n, int16_t *pcm,
                       int8_t *dbg_ip, int8_t *dbg_qp, float *dbg_theta, float *dbg_dtheta,
                       float *dbg_deemph, int16_t *dbg_w)
{
    size_t i, m;
    float k;
    tables_init();
    split_rails(iq, n, c->ri, c->rq);
    q15_run(&w->pre_i, c->ri, n, c->a16, c->work);
    q15_run(&w->pre_q, c->rq, n, c->b16, c->work);
    k = w->gain / 75000;
    k *= 32767;
    for (i = 0; i < n; i++) {
        int8_t ip = (int8_t)c->a16[i]; /* :393,:397 narrow back to int8 */
        int8_t qp = (int8_t)c->b16[i];
        uint8_t xi = (uint8_t)((uint8_t)ip + 128); /* :458-459 */
        uint8_t yi = (uint8_t)((uint8_t)qp + 128);
        float theta = g_atan2_lut[yi][xi];
        float d = theta - w->theta_prev;
        float y;
        d = wrap_delta(d);
        y = f32_iir_step(&w->deemph, k * d);
        w->theta_prev = theta;
        c->c16[i] = iqo_cast_i16(y); /* :536 */
        if (dbg_ip) { dbg_ip[i] = ip; dbg_qp[i] = qp; dbg_theta[i] = theta;
                      dbg_dtheta[i] = d; dbg_deemph[i] = y; dbg_w[i] = c->c16[i]; }
    }
    m = q15_run(&w->d1, c->c16, n, c->a16, c->work);
    m = q15_run(&w->d2, c->a16, m, c->b16, c->work);
    m = q15_run(&w->d3, c->b16, m, pcm, c->work);
    return m;
}

static void fm_init(fm_t *f)
{
    /* FmDemodulator.cc:113-122: the literals -1/16 and 1/16 are integer
     * divisions, so the "differentiator" is {0,0,1,0,-1,0,0}. */
    static const float diff[7] = {-1 / 16, 0, 1, 0, -1, 0, 1 / 16};
    q15_init(&f->tuner_i, T_FM_TUNER, 4);
    q15_init(&f->tuner_q, T_FM_TUNER, 4);
    q15_init(&f->post, T_FM_POST, 4);
    q15_init(&f->audio, T_AUDIO40, 2);
    f32_fir_init(&f->diff, diff, 7);
    f->gain = 64000 / (2 * M_PI); /* FmDemodulator.cc:158 */
}

/* FmDemodulator.cc resetDemodulator: the four decimators and the differentiator. */
static void fm_reset(fm_t *f)
{
    q15_reset(&f->tuner_i); q15_reset(&f->tuner_q);
    q15_reset(&f->post); q15_reset(&f->audio);
    memset(f->diff.x, 0, sizeof(f->diff.x));
}

/* FmDemodulator.cc:376-423, :460-504, :526-560 */
static size_t fm_run(iqo_chain *c, fm_t *f, const int8_t *iq, size_t n, int16_t *pcm)
{
    size_t i, m;
    float k;
    split_rails(iq, n, c->ri, c->rq);
    m = q15_run(&f->tuner_i, c->ri, n, c->a16, c->work);
    m = q15_run(&f->tuner_q, c->rq, n, c->b16, c->work);
    k = f->gain / 15000;
    k *= 32767;
    for (i = 0; i < m; i++) {
        float theta = atan2((double)c->b16[i], (double)c->a16[i]);
        float d = f32_fir_step(&f->diff, theta);
        d = wrap_delta(d);
        c->c16[i] = iqo_cast_i16(k * d);
    }
    m = q15_run(&f->post, c->c16, m, c->a16, c->work);
    m = q15_run(&f->audio, c->a16, m, pcm, c->work);
    return m;
}

static void front32_init(q15_filter *s1i, q15_filter *s1q, q15_filter *s2i, q15_filter *s2q,
                         q15_filter *s3i, q15_filter *s3q)
{
    q15_init(s1i, T_AM_S1, 4); q15_init(s1q, T_AM_S1, 4);
    q15_init(s2i, T_AM_S2, 4); q15_init(s2q, T_AM_S2, 4);
    q15_init(s3i, T_AM_S3, 2); q15_init(s3q, T_AM_S3, 2);
}

static void am_init(am_t *a)
{
    front32_init(&a->s1i, &a->s1q, &a->s2i, &a->s2q, &a->s3i, &a->s3q);
    f32_iir_init(&a->dc, B_DCBLOCK, 2, A_DCBLOCK, 1);
    a->gain = 300; /* AmDemodulator.cc:104 */
}

static void am_reset(am_t *a)
{
    q15_reset(&a->s1i); q15_reset(&a->s1q); q15_reset(&a->s2i);
    q15_reset(&a->s2q); q15_reset(&a->s3i); q15_reset(&a->s3q);
    f32_iir_reset(&a->dc);
}

/* The shared /32 front end: AmDemodulator.cc:339-408, SsbDemodulator.cc:462-529 */
static size_t front32_run(iqo_chain *c, q15_filter *s1i, q15_filter *s1q, q15_filter *s2i,
                          q15_filter *s2q, q15_filter *s3i, q15_filter *s3q,
                          const int8_t *iq, size_t n, int16_t *i8k, int16_t *q8k)
{
    size_t m;
    split_rails(iq, n, c->ri, c->rq);
    m = q15_run(s1i, c->ri, n, c->a16, c->work);
    m = q15_run(s2i, c->a16, m, c->b16, c->work);
    m = q15_run(s3i, c->b16, m, i8k, c->work);
    m = q15_run(s1q, c->rq, n, c->a16, c->work);
    m = q15_run(s2q, c->a16, m, c->b16, c->work);
    m = q15_run(s3q, c->b16, m, q8k, c->work);
    return m;
}

/* AmDemodulator.cc:434-471 */
static size_t am_run(iqo_chain *c, am_t *a, const int8_t *iq, size_t n, int16_t *pcm)
{
    size_t i, m;
    m = front32_run(c, &a->s1i, &a->s1q, &a->s2i, &a->s2q, &a->s3i, &a->s3q, iq, n,
                    c->c16, c->d16);
    for (i = 0; i < m; i++) {
        int16_t im = (int16_t)abs((int)c->c16[i]);
        int16_t qm = (int16_t)abs((int)c->d16[i]);
        int16_t in = (im > qm) ? (int16_t)(im + (qm >> 1)) : (int16_t)(qm + (im >> 1));
        float y = f32_iir_step(&a->dc, (float)in);
        pcm[i] = iqo_cast_i16(a->gain * y);
    }
    return m;
}

static void ssb_init(ssb_t *s)
{
    front32_init(&s->s1i, &s->s1q, &s->s2i, &s->s2q, &s->s3i, &s->s3q);
    q15_init(&s->delay, T_SSB_DELAY, 1);
    q15_init(&s->hilbert, T_SSB_HILBERT, 1);
    f32_iir_init(&s->dc, B_DCBLOCK, 2, A_DCBLOCK, 1);
    s->gain = 300; /* SsbDemodulator.cc:147 */
    s->lsb = 1;    /* SsbDemodulator.cc:144 */
}

static void ssb_reset(ssb_t *s)
{
    q15_reset(&s->s1i); q15_reset(&s->s1q); q15_reset(&s->s2i);
    q15_reset(&s->s2q); q15_reset(&s->s3i); q15_reset(&s->s3q);
    q15_reset(&s->delay); q15_reset(&s->hilbert);
    f32_iir_reset(&s->dc);
}

/* SsbDemodulator.cc:563-598 */
static size_t ssb_run(iqo_chain *c, ssb_t *s, const int8_t *iq, size_t n, int16_t *pcm)
{
    size_t i, m;
    m = front32_run(c, &s->s1i, &s->s1q, &s->s2i, &s->s2q, &s->s3i, &s->s3q, iq, n,
                    c->c16, c->d16);
    q15_run(&s->delay, c->c16, m, c->a16, c->work);
    q15_run(&s->hilbert, c->d16, m, c->b16, c->work);
    for (i = 0; i < m; i++) {
        float o = s->lsb ? (float)(c->a16[i] - c->b16[i]) : (float)(c->a16[i] + c->b16[i]);
        float y = f32_iir_step(&s->dc, o);
        pcm[i] = iqo_cast_i16(s->gain * y);
    }
    return m;
}

/* ------------------------------------------------------------------------ */
/* Chain = IqDataProcessor                                                    */
/* ------------------------------------------------------------------------ */
/* ------------------------------------------------------------------------ */
/* Float / int16 resamplers (Filters/Decimator.cc, Interpolator.cc,           */
/* Int16/Interpolator_int16.cc): streams from the zero state                 */
/* ------------------------------------------------------------------------ */
/* Decimator::decimate :283-321 -> filterData :176-214: y = 0; y = y + h[k]*x[n-k], k ascending, binary32;
 * one output per `factor` inputs, taken when the last of them arrives. */
long iqo_decimate_f32(const float *h, int length, int factor, const float *in, size_t n, float *out)
{
    size_t m = 0, i;
    for (i = (size_t)factor - 1; i < n; i += (size_t)factor) {
        float y = 0;
        int k;
        for (k = 0; k < length; k++) {
            const float x = (size_t)k <= i ? in[i - (size_t)k] : 0.0f;
            y = y + (h[k] * x);
        }
        out[m++] = y;
    }
    return (long)m;
}

/* Interpolator::interpolate :340-364 with createPolyphaseCoefficients :263-300: sub-filter i holds
 * h[i], h[i+L], ...; q = length / L taps each (integer division); L outputs per input. */
void iqo_interpolate_f32(const float *h, int length, int factor, const float *in, size_t n, float *out)
{
    const int q = length / factor;
    size_t i;
    for (i = 0; i < n; i++) {
        int p, k;
        for (p = 0; p < factor; p++) {
            float y = 0;
            for (k = 0; k < q; k++) {
                const float x = (size_t)k <= i ? in[i - (size_t)k] : 0.0f;
                y = y + (h[p + k * factor] * x);
            }
            out[i * (size_t)factor + (size_t)p] = y;
        }
    }
}

/* Interpolator_int16: taps quantised like every Q15 filter (createPolyphaseCoefficients :270-296), then the
 * Q15 accumulator with its per-MAC clamp (filterData :203-246). */
void iqo_interpolate_q15(const float *h, int length, int factor, const int16_t *in, size_t n, int16_t *out)
{
    const int q = length / factor;
    int16_t hq[1024];
    size_t i;
    if (length > 1024) return;
    iqo_quantize_taps(h, length, hq);
    for (i = 0; i < n; i++) {
        int p, k;
        for (p = 0; p < factor; p++) {
            int32_t acc = 1 << 14;
            for (k = 0; k < q; k++) {
                const int32_t x = (size_t)k <= i ? in[i - (size_t)k] : 0;
                acc = acc + ((int32_t)hq[p + k * factor] * x);
                if (acc > 0x3fffffff) acc = 0x3fffffff;
                if (acc < -0x40000000) acc = -0x40000000;
            }
            out[i * (size_t)factor + (size_t)p] = (int16_t)(acc >> 15);
        }
    }
}

/* ------------------------------------------------------------------------ */
/* AutomaticGainControl (src_diags/AutomaticGainControl.cc)                   */
/* ------------------------------------------------------------------------ */
#define AGC_MAX_GAIN 46 /* MAX_ADJUSTIBLE_GAIN, AutomaticGainControl.cc:23 */

/* Constructor defaults, AutomaticGainControl.cc:113-189 (operating point -12: Radio.cc:184) */
static void agc_init(iqo_agc *a)
{
    a->type = IQO_AGC_HARRIS;
    a->deadband_db = 1;
    a->signal_magnitude = 64;
    a->enabled = 0;
    a->blanking_counter = 0;
    a->blanking_limit = 1;
    a->gain_was_adjusted = 0;
    a->if_gain_db = 24;
    a->normalized_level_dbfs = -24;
    a->filtered_if_gain_db = 24;
    a->alpha = 0.8;
    a->operating_point_dbfs = -12;
}

static void agc_reset_blanking(iqo_agc *a) /* :625-634 */
{
    a->blanking_counter = 0;
    a->gain_was_adjusted = 0;
}

/* runLowpass :743-889 / runHarris :935-1062.  `gain` is the radio's IF gain
 * (Radio::receiveIfGainInDb == radio_adjustableReceiveGainInDb); returns it updated. */
static uint32_t agc_adjust(iqo_agc *a, uint32_t magnitude, uint32_t gain)
{
    int32_t signal_dbfs, error;
    a->signal_magnitude = magnitude;
    signal_dbfs = iqo_dbfs(magnitude);
    a->normalized_level_dbfs = (int32_t)((uint32_t)signal_dbfs - a->if_gain_db);
    error = a->operating_point_dbfs - signal_dbfs;
    if (a->if_gain_db == AGC_MAX_GAIN) {
        if (error > 0) error = 0;
    } else if (a->if_gain_db == 0) {
        if (error < 0) error = 0;
    }
    if (abs(error) <= a->deadband_db) error = 0;
    if (a->type == IQO_AGC_LOWPASS) {
        int32_t adjusted = (int32_t)(a->if_gain_db + (uint32_t)error);
        a->filtered_if_gain_db = (a->alpha * (float)adjusted) + ((1 - a->alpha) * a->filtered_if_gain_db);
    } else {
        a->filtered_if_gain_db = a->filtered_if_gain_db + (a->alpha * (float)error);
    }
    if (a->filtered_if_gain_db > AGC_MAX_GAIN) a->filtered_if_gain_db = AGC_MAX_GAIN;
    else if (a->filtered_if_gain_db < 0) a->filtered_if_gain_db = 0;
    a->if_gain_db = (uint32_t)a->filtered_if_gain_db;
    if (error != 0) {
        gain = a->if_gain_db;      /* Radio::setReceiveIfGainInDb(0, ifGainInDb) */
        a->gain_was_adjusted = 1;
    }
    return gain;
}

/* run() :663-741: follows external gain changes, blanks after an adjustment. */
uint32_t iqo_agc_run(iqo_agc *a, uint32_t magnitude, uint32_t gain)
{
    int allowed = 0;
    if (a->if_gain_db != gain) a->if_gain_db = gain;
    if (a->gain_was_adjusted) {
        if (a->blanking_counter < a->blanking_limit) {
            a->blanking_counter++;
        } else {
            agc_reset_blanking(a);
            allowed = 1;
        }
    } else {
        allowed = 1;
    }
    if (allowed) gain = agc_adjust(a, magnitude, gain);
    return gain;
}

iqo_agc *iqo_agc_of(iqo_chain *c) { return &c->agc; }
/* signalMagnitudeCallback, AutomaticGainControl.cc:47-64 */
void iqo_agc_feed(iqo_chain *c, uint32_t magnitude)
{
    if (c->agc.enabled) c->rx_gain_db = iqo_agc_run(&c->agc, magnitude, c->rx_gain_db);
}
uint32_t iqo_get_rx_gain_db(const iqo_chain *c) { return c->rx_gain_db; }

int iqo_agc_set_type(iqo_chain *c, uint32_t type) /* :287-320 */
{
    if (type != IQO_AGC_LOWPASS && type != IQO_AGC_HARRIS) return 0;
    c->agc.type = type;
    return 1;
}
int iqo_agc_set_deadband(iqo_chain *c, uint32_t db) /* :351-369 */
{
    if (db > 10) return 0;
    c->agc.deadband_db = (int32_t)db;
    return 1;
}
int iqo_agc_set_blanking_limit(iqo_chain *c, uint32_t limit) /* :399-420 */
{
    if (limit > 10) return 0;
    c->agc.blanking_limit = limit;
    agc_reset_blanking(&c->agc);
    return 1;
}
void iqo_agc_set_operating_point(iqo_chain *c, int32_t dbfs) { c->agc.operating_point_dbfs = dbfs; } /* :440-448 */
int iqo_agc_set_filter_coefficient(iqo_chain *c, float coefficient) /* :475-493, double compares */
{
    if ((coefficient >= 0.001) && (coefficient < 0.999)) {
        c->agc.alpha = coefficient;
        return 1;
    }
    return 0;
}
int iqo_agc_enable(iqo_chain *c, int on) /* enable :516-548 (the radio is receiving), disable :571-595 */
{
    if (on) {
        if (c->agc.enabled) return 0;
        agc_reset_blanking(&c->agc);
        c->agc.enabled = 1;
        return 1;
    }
    if (!c->agc.enabled) return 0;
    c->agc.enabled = 0;
    return 1;
}

/* ------------------------------------------------------------------------ */
/* FrequencyScanner (src_diags/FrequencyScanner.cc)                           */
/* ------------------------------------------------------------------------ */
static void scanner_init(iqo_scanner *s) /* constructor :96-131 */
{
    s->start_hz = 162550000;
    s->end_hz = 162550000;
    s->increment_hz = 0;
    s->current_hz = s->start_hz;
    s->new_configuration = 0;
    s->scanning = 0;
    s->tuned_hz = 0;
    s->tune_count = 0;
}

iqo_scanner *iqo_scanner_of(iqo_chain *c) { return &c->scanner; }

int iqo_scanner_set_parameters(iqo_chain *c, uint64_t start_hz, uint64_t end_hz, uint64_t increment_hz) /* :190-218 */
{
    iqo_scanner *s = &c->scanner;
    if (s->scanning) return 0;
    s->start_hz = start_hz;
    s->end_hz = end_hz;
    s->increment_hz = increment_hz;
    s->new_configuration = 1;
    return 1;
}

static void scanner_tune(iqo_scanner *s) /* Radio::setReceiveFrequency(currentFrequencyInHertz) */
{
    s->tuned_hz = s->current_hz;
    s->tune_count++;
}

int iqo_scanner_start(iqo_chain *c) /* :240-270 */
{
    iqo_scanner *s = &c->scanner;
    if (s->scanning) return 0;
    if (s->new_configuration) {
        s->current_hz = s->end_hz;
        scanner_tune(s);
        s->new_configuration = 0;
    }
    s->scanning = 1;
    return 1;
}

int iqo_scanner_stop(iqo_chain *c) /* :292-310 */
{
    if (!c->scanner.scanning) return 0;
    c->scanner.scanning = 0;
    return 1;
}

/* signalStateCallback :47-62 -> run :378-404 */
void iqo_scanner_feed(iqo_chain *c, int signal_present)
{
    iqo_scanner *s = &c->scanner;
    if (!s->scanning || signal_present) return;
    s->current_hz = s->current_hz + s->increment_hz;
    if (s->current_hz > s->end_hz) s->current_hz = s->start_hz;
    scanner_tune(s);
}

iqo_chain *iqo_create(void)
{
    iqo_chain *c = (iqo_chain *)calloc(1, sizeof(*c));
    if (!c) return 0;
    tables_init();
    c->mode = IQO_NONE;       /* IqDataProcessor.cc:38 */
    c->rotation = 1;
    c->threshold = -200;      /* IqDataProcessor.cc:41 */
    c->rx_gain_db = 24;       /* Radio.cc:325-328 */
    c->tracking = 0;
    agc_init(&c->agc);
    scanner_init(&c->scanner);
    wbfm_init(&c->wbfm);
    fm_init(&c->fm);
    am_init(&c->am);
    ssb_init(&c->ssb);
    return c;
}

void iqo_destroy(iqo_chain *c) { free(c); }

void iqo_reset(iqo_chain *c)
{
    wbfm_reset(&c->wbfm);
    fm_reset(&c->fm);
    am_reset(&c->am);
    ssb_reset(&c->ssb);
}

/* one demodulator's resetDemodulator(): which = 1 AM, 2 FM, 3 WBFM, 4 SSB */
void iqo_reset_demod(iqo_chain *c, int which)
{
    switch (which) {
    case 1: am_reset(&c->am); break;
    case 2: fm_reset(&c->fm); break;
    case 3: wbfm_reset(&c->wbfm); break;
    case 4: ssb_reset(&c->ssb); break;
    default: break;
    }
}

/* IqDataProcessor.cc:236-262 */
void iqo_set_mode(iqo_chain *c, int mode)
{
    c->mode = mode;
    if (mode == IQO_LSB) c->ssb.lsb = 1;
    if (mode == IQO_USB) c->ssb.lsb = 0;
}

void iqo_set_gain(iqo_chain *c, int which, float gain)
{
    switch (which) {
    case 1: c->am.gain = gain; break;
    case 2: c->fm.gain = gain; break;
    case 3: c->wbfm.gain = gain; break;
    case 4: c->ssb.gain = gain; break;
    default: break;
    }
}

void iqo_set_squelch(iqo_chain *c, int32_t threshold) { c->threshold = threshold; }
void iqo_set_rx_gain_db(iqo_chain *c, uint32_t g) { c->rx_gain_db = g; }
void iqo_set_rotation(iqo_chain *c, int r) { c->rotation = r; }

static size_t run_demod(iqo_chain *c, int mode, const int8_t *s8, size_t n, int16_t *pcm)
{
    switch (mode) {
    case IQO_AM: return am_run(c, &c->am, s8, n, pcm);
    case IQO_FM: return fm_run(c, &c->fm, s8, n, pcm);
    case IQO_WBFM: return wbfm_run(c, &c->wbfm, s8, n, pcm, 0, 0, 0, 0, 0, 0);
    case IQO_LSB: case IQO_USB: return ssb_run(c, &c->ssb, s8, n, pcm);
    default: return 0;
    }
}

/* IqDataProcessor.cc:722-840 */
long iqo_accept(iqo_chain *c, const uint8_t *iq, size_t bytes, int16_t *pcm, size_t cap,
                uint32_t *magnitude, uint8_t *allowed_out)
{
    size_t i, n = bytes / 2, produced;
    uint32_t mag;
    int32_t dbfs;
    int present, allowed;
    static int16_t tmp[MAX_BLOCK_SAMPLES / 32 + 8];
    if (bytes > 2 * MAX_BLOCK_SAMPLES || (bytes % 8) != 0) return -1;
    for (i = 0; i < bytes; i++) c->s8[i] = (int8_t)(uint8_t)(iq[i] - 128); /* :735-738 */
    iqo_rotate(c->s8, bytes, c->rotation);                                 /* :749 */
    /* Squelch::run, Squelch.cc:227-273 */
    mag = iqo_block_magnitude(c->s8, bytes);
    dbfs = iqo_dbfs(mag);
    dbfs = (int32_t)((uint32_t)dbfs - c->rx_gain_db); /* SignalDetector.cc:262 */
    present = dbfs >= c->threshold;
    /* SignalTracker.cc:104-145: allowed for START / PRESENT / END events */
    allowed = present || c->tracking;
    c->tracking = present;
    if (magnitude) *magnitude = mag;
    if (allowed_out) *allowed_out = (uint8_t)allowed;
    iqo_scanner_feed(c, allowed); /* signalCallbackPtr(signalAllowed), :771-775 -> FrequencyScanner.cc:47-62 */
    /* signalMagnitudeCallback (:781-790 -> AutomaticGainControl.cc:47-64): after the squelch, before the
     * demodulator; a gain change is seen by the next block's squelch */
    if (c->agc.enabled) c->rx_gain_db = iqo_agc_run(&c->agc, mag, c->rx_gain_db);
    if (!allowed) return 0; /* :793 */
    produced = run_demod(c, c->mode, c->s8, n, tmp);
    for (i = 0; i < produced && i < cap; i++) pcm[i] = tmp[i];
    return (long)produced;
}

long iqo_accept_stream(iqo_chain *c, const uint8_t *iq, size_t total, size_t block_bytes,
                       int16_t *pcm, size_t cap, uint32_t *magnitude, uint8_t *allowed)
{
    size_t off, produced = 0, block = 0;
    for (off = 0; off < total; off += block_bytes, block++) {
        size_t n = (total - off < block_bytes) ? (total - off) : block_bytes;
        long got = iqo_accept(c, iq + off, n, pcm + produced, cap > produced ? cap - produced : 0,
                              magnitude ? magnitude + block : 0, allowed ? allowed + block : 0);
        if (got < 0) return got;
        produced += (size_t)got;
    }
    return (long)produced;
}

long iqo_demod_accept(iqo_chain *c, int mode, const int8_t *iq, size_t bytes,
                      int16_t *pcm, size_t cap)
{
    static int16_t tmp[MAX_BLOCK_SAMPLES / 32 + 8];
    size_t i, produced;
    if (bytes > 2 * MAX_BLOCK_SAMPLES) return -1;
    if (mode == IQO_LSB) c->ssb.lsb = 1;
    if (mode == IQO_USB) c->ssb.lsb = 0;
    produced = run_demod(c, mode, iq, bytes / 2, tmp);
    for (i = 0; i < produced && i < cap; i++) pcm[i] = tmp[i];
    return (long)produced;
}

/* ------------------------------------------------------------------------ */
/* Primitive entry points for known-answer tests                              */
/* ------------------------------------------------------------------------ */
long iqo_decimate_q15(const float *h, int length, int factor,
                      const int16_t *in, size_t n, int16_t *out)
{
    q15_filter f;
    int16_t *work;
    size_t m;
    if (length > QMAX) return -1;
    work = (int16_t *)malloc((n + QMAX) * sizeof(int16_t));
    if (!work) return -1;
    q15_init_taps(&f, h, length, factor);
    m = q15_run(&f, in, n, out, work);
    free(work);
    return (long)m;
}

void iqo_fir_q15(const float *h, int length, const int16_t *in, size_t n, int16_t *out)
{
    (void)iqo_decimate_q15(h, length, 1, in, n, out);
}

void iqo_fir_f32(const float *h, int length, const float *in, size_t n, float *out)
{
    f32_fir f;
    size_t i;
    f32_fir_init(&f, h, length);
    for (i = 0; i < n; i++) out[i] = f32_fir_step(&f, in[i]);
}

void iqo_iir_f32(const float *b, int nb, const float *a, int na,
                 const float *in, size_t n, float *out)
{
    f32_iir f;
    size_t i;
    f32_iir_init(&f, b, nb, a, na);
    for (i = 0; i < n; i++) out[i] = f32_iir_step(&f, in[i]);
}

int iqo_get_taps_f32(int which, float *h) { return expand_taps(which, h); }

int iqo_get_taps_q15(int which, int16_t *hq)
{
    float h[QMAX];
    int n = expand_taps(which, h);
    iqo_quantize_taps(h, n, hq);
    return n;
}

void iqo_wbfm_stages(const int8_t *rot, size_t n, float gain, int8_t *ip, int8_t *qp,
                     float *theta, float *dtheta, float *deemph, int16_t *w)
{
    iqo_chain *c = iqo_create();
    int16_t pcm[MAX_BLOCK_SAMPLES / 32 + 8];
    size_t off;
    c->wbfm.gain = gain;
    for (off = 0; off < n; off += MAX_BLOCK_SAMPLES) {
        size_t m = (n - off < MAX_BLOCK_SAMPLES) ? (n - off) : MAX_BLOCK_SAMPLES;
        wbfm_run(c, &c->wbfm, rot + 2 * off, m, pcm, ip + off, qp + off, theta + off,
                 dtheta + off, deemph + off, w + off);
    }
    iqo_destroy(c);
}
