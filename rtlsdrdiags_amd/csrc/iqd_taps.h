// Filter designs of the four demodulator chains as the reference states them (float
// coefficients; the reference quantises them to Q15 when it constructs its filters,
// Decimator_int16.cc:55-63 / FirFilter_int16.cc:46-54, and so does build_consts()).
// Sources: WbFmDemodulator.cc:17-123, FmDemodulator.cc:14-122, AmDemodulator.cc:14-68,
// SsbDemodulator.cc:14-107.  Every design is linear phase; the tables list all taps.
#pragma once

namespace iqd {
namespace taps {

constexpr float WBFM_PRE[16] = {
    -0.0157211f, -0.0325959f, 0.0092996f, 0.0621217f, -0.0148595f, -0.0989456f, 0.1182989f, 0.4862333f,
    0.4862333f, 0.1182989f, -0.0989456f, -0.0148595f, 0.0621217f, 0.0092996f, -0.0325959f, -0.0157211f};

constexpr float WBFM_D1[8] = {0.0243699f, 0.0769537f, 0.1463572f, 0.1967096f,
                              0.1967096f, 0.1463572f, 0.0769537f, 0.0243699f};

// second /4 stage of WBFM and the post-demodulation /4 stage of FM (same design)
constexpr float POST12[12] = {0.0022977f, 0.0237042f, 0.0605386f, 0.1127073f, 0.1645167f, 0.1971107f,
                              0.1971107f, 0.1645167f, 0.1127073f, 0.0605386f, 0.0237042f, 0.0022977f};

// 16 kS/s -> 8 kS/s audio decimator shared by WBFM and FM
constexpr float AUDIO40[40] = {
    0.0015969f, -0.0111080f, -0.0270501f, -0.0265610f, -0.0023190f, 0.0180618f, 0.0065495f, -0.0183409f,
    -0.0133345f, 0.0184489f, 0.0230891f, -0.0161248f, -0.0363745f, 0.0091343f, 0.0550219f, 0.0070312f,
    -0.0862280f, -0.0497761f, 0.1793543f, 0.4145808f, 0.4145808f, 0.1793543f, -0.0497761f, -0.0862280f,
    0.0070312f, 0.0550219f, 0.0091343f, -0.0363745f, -0.0161248f, 0.0230891f, 0.0184489f, -0.0133345f,
    -0.0183409f, 0.0065495f, 0.0180618f, -0.0023190f, -0.0265610f, -0.0270501f, -0.0111080f, 0.0015969f};

constexpr float FM_TUNER[32] = {
    0.0041331f, 0.0054174f, 0.0076016f, 0.0115481f, 0.0151685f, 0.0203192f, 0.0251608f, 0.0311322f,
    0.0366372f, 0.0427168f, 0.0480527f, 0.0533425f, 0.0575831f, 0.0611914f, 0.0635413f, 0.0648239f,
    0.0648239f, 0.0635413f, 0.0611914f, 0.0575831f, 0.0533425f, 0.0480527f, 0.0427168f, 0.0366372f,
    0.0311322f, 0.0251608f, 0.0203192f, 0.0151685f, 0.0115481f, 0.0076016f, 0.0054174f, 0.0041331f};

// AM and SSB share the three-stage /32 front end
constexpr float AM_S1[8] = {0.0242683f, 0.0766338f, 0.1457589f, 0.1959036f,
                            0.1959036f, 0.1457589f, 0.0766338f, 0.0242683f};
constexpr float AM_S2[12] = {0.0057496f, 0.0263853f, 0.0605301f, 0.1074406f, 0.1523486f, 0.1804951f,
                             0.1804951f, 0.1523486f, 0.1074406f, 0.0605301f, 0.0263853f, 0.0057496f};
constexpr float AM_S3[16] = {0.0116487f, 0.0152694f, -0.0109804f, -0.0611915f, -0.0736143f, 0.0187617f,
                             0.1988190f, 0.3481364f, 0.3481364f, 0.1988190f, 0.0187617f, -0.0736143f,
                             -0.0611915f, -0.0109804f, 0.0152694f, 0.0116487f};

// SSB phasing method: a 15-sample delay for I (the tap 1.0 quantises to -32768, so the
// "delay line" also inverts) and a 31-tap windowed Hilbert transformer for Q
constexpr float SSB_DELAY[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1};
constexpr float SSB_HILBERT[31] = {
    -0.0033953f, 0, -0.0058652f, 0, -0.0134385f, 0, -0.0281423f, 0, -0.0534836f, 0, -0.0980394f, 0,
    -0.1935638f, 0, -0.6302204f, 0, 0.6302204f, 0, 0.1935638f, 0, 0.0980394f, 0, 0.0534836f, 0,
    0.0281423f, 0, 0.0134385f, 0, 0.0058652f, 0, 0.0033953f};

constexpr float DEEMPH_B0 = 0.0253863f;   // both numerator taps
constexpr float DEEMPH_A1 = -0.9492274f;
constexpr float DCBLOCK_A1 = -0.95f;      // numerator {1, -1}

// quantize_q15() (iqd_host.cpp: h * 32768, roundf, int16) at compile time: the designs are fixed, so a kernel can carry its taps
// as literals of the instructions instead of in scalar registers (|h * 32768| < 2^15: the sum with 0.5 is exact).
// tests/test_host_logic.py compares with what the host builds.
constexpr int q15(float h)
{
    const float s = h * 32768.0f;
    return (int)(s + (s >= 0.f ? 0.5f : -0.5f));
}
constexpr unsigned pair16(int lo, int hi) { return ((unsigned)lo & 0xffffu) | ((unsigned)hi << 16); }
// The WBFM streaming kernel's IIR lanes (build_stream_taps(): StreamArgs::d1p2, p12p, a40p) and the FM pipeline's consumer lanes.
struct StreamTaps {
    unsigned d1p2[4], p12p[6], a40p[20];
    constexpr StreamTaps() : d1p2{}, p12p{}, a40p{}
    {
        for (int q = 0; q < 4; q++) d1p2[q] = pair16(2 * q15(WBFM_D1[7 - 2 * q]), 2 * q15(WBFM_D1[6 - 2 * q]));
        for (int q = 0; q < 6; q++) p12p[q] = pair16(q15(POST12[2 * q + 1]), q15(POST12[2 * q]));
        for (int q = 0; q < 20; q++) a40p[q] = pair16(q15(AUDIO40[2 * q + 1]), q15(AUDIO40[2 * q]));
    }
};
constexpr StreamTaps STREAM_TAPS{};

}  // namespace taps
}  // namespace iqd
