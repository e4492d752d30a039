#include "iqd_host.h"
#include <stdlib.h>

#include <math.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "iqd_taps.h"
#include "iqd_prims.h"
#include "iqd_stream.h"

namespace iqd {

int16_t host_cast_i16(float f)
{
    int32_t wide;
    if (f >= -2147483648.0f && f < 2147483648.0f) wide = (int32_t)f;
    else wide = INT32_MIN;
    return (int16_t)(uint16_t)((uint32_t)wide & 0xffffu);
}

void quantize_q15(const float *h, int n, int16_t *hq)
{
    for (int i = 0; i < n; i++) {
        float scaled = h[i] * 32768;
        scaled = roundf(scaled);
        hq[i] = host_cast_i16(scaled);
    }
}

// Packs Q15 taps for v_dot4_i32_i8 over a window stored oldest sample first:
// window byte b multiplies tap h[L-1-b]; each tap is split as h = 256*hi + lo with lo in
// [-128,127], so that sum(h*x) = sum(lo*x) + 256*sum(hi*x).
static void pack_dot4(const int16_t *hq, int L, int32_t *lo, int32_t *hi)
{
    for (int q = 0; q < L / 4; q++) {
        uint32_t l = 0, h = 0;
        for (int b = 0; b < 4; b++) {
            const int tap = hq[L - 1 - (4 * q + b)];
            const int8_t tl = (int8_t)(tap & 0xff);
            const int th = (tap - tl) / 256;
            l |= (uint32_t)(uint8_t)tl << (8 * b);
            h |= (uint32_t)(uint8_t)(int8_t)th << (8 * b);
        }
        lo[q] = (int32_t)l;
        hi[q] = (int32_t)h;
    }
}

// The WBFM pre-demod FIR wants its accumulator doubled, so that the 8 bits the atan2 table is
// indexed with (bits 15..22 of the Q15 sum) land on a byte boundary (bits 16..23):
// h = 128*hi + lo with lo in [-64,63];  2*sum(h*x) = sum((2*lo)*x) + 256*sum(hi*x).
static void pack_dot4_doubled(const int16_t *hq, int L, int32_t *lo2, int32_t *hi)
{
    for (int q = 0; q < L / 4; q++) {
        uint32_t l = 0, h = 0;
        for (int b = 0; b < 4; b++) {
            const int tap = hq[L - 1 - (4 * q + b)];
            int tl = tap & 0x7f;
            if (tl >= 64) tl -= 128;
            const int th = (tap - tl) / 128;     // |tap| <= 15933 -> |th| <= 125
            l |= (uint32_t)(uint8_t)(int8_t)(2 * tl) << (8 * b);
            h |= (uint32_t)(uint8_t)(int8_t)th << (8 * b);
        }
        lo2[q] = (int32_t)l;
        hi[q] = (int32_t)h;
    }
}

void build_consts(Consts &c)
{
    memset(&c, 0, sizeof(c));
    int16_t pre[16];
    quantize_q15(taps::WBFM_PRE, 16, pre);
    pack_dot4_doubled(pre, 16, c.pre_lo, c.pre_hi);
    quantize_q15(taps::WBFM_D1, 8, c.wbfm_d1);
    quantize_q15(taps::POST12, 12, c.post12);
    quantize_q15(taps::AUDIO40, 40, c.audio40);
    quantize_q15(taps::FM_TUNER, 32, c.fm_tuner);
    pack_dot4(c.fm_tuner, 32, c.fm_tuner_lo, c.fm_tuner_hi);
    quantize_q15(taps::AM_S1, 8, c.am_s1);
    pack_dot4(c.am_s1, 8, c.am_s1_lo, c.am_s1_hi);
    quantize_q15(taps::AM_S2, 12, c.am_s2);
    quantize_q15(taps::AM_S3, 16, c.am_s3);
    quantize_q15(taps::SSB_DELAY, 16, c.ssb_delay);
    quantize_q15(taps::SSB_HILBERT, 31, c.ssb_hilbert);
    // DbfsCalculator.cc:36-68 (C++: log10((float)i) is the float overload)
    for (uint32_t i = 1; i <= 256; i++) {
        float level = 20 * log10f((float)i);
        c.db_table[i] = (int32_t)level;
    }
    c.db_table[0] = c.db_table[1];
    c.deemph_b0 = taps::DEEMPH_B0;
    c.deemph_a1 = taps::DEEMPH_A1;
    const double cc = -(double)taps::DEEMPH_A1;
    c.deemph_c = (float)cc;
    c.deemph_c16 = (float)pow(cc, 16.0);
    c.deemph_c127 = (float)pow(cc, 127.0);
    c.deemph_c128 = (float)pow(cc, 128.0);
    c.deemph_cinv = (float)(1.0 / cc);
    c.dc_a1 = taps::DCBLOCK_A1;
    c.dc_cseg = (float)pow(-(double)taps::DCBLOCK_A1, 32.0);
}

void build_atan2_lut(std::vector<float> &lut)
{
    lut.resize(256 * 256);
    for (int x = 0; x < 256; x++)
        for (int y = 0; y < 256; y++)
            lut[lut_index((uint32_t)(y * 256 + x))] = (float)atan2((double)y - 128, (double)x - 128);
}

void build_fm_lut(std::vector<float> &lut)
{
    lut.resize((size_t)FM_LUT_W * FM_LUT_W);
    for (int q = -FM_LUT_R; q <= FM_LUT_R; q++)
        for (int i = -FM_LUT_R; i <= FM_LUT_R; i++)
            lut[(size_t)(q + FM_LUT_R) * FM_LUT_W + (i + FM_LUT_R)] = (float)atan2((double)q, (double)i);
}

void default_params(ChanParams &p)
{
    memset(&p, 0, sizeof(p));
    p.mode = 0;               // IqDataProcessor.cc:38
    p.threshold = -200;       // IqDataProcessor.cc:41
    p.rx_gain_db = 24;        // Radio.cc:325-328
    p.rotation = 1;           // IqDataProcessor.cc:749
    p.ssb_lsb = 1;            // SsbDemodulator.cc:144
    p.gain[FAM_AM] = 300;                       // AmDemodulator.cc:104
    p.gain[FAM_FM] = 64000 / (2 * M_PI);        // FmDemodulator.cc:158
    p.gain[FAM_WBFM] = 256000 / (2 * M_PI);     // WbFmDemodulator.cc:173
    p.gain[FAM_SSB] = 300;                      // SsbDemodulator.cc:147
    derive_params(p);
}

void derive_params(ChanParams &p)
{
    // WbFmDemodulator.cc:447-450 / FmDemodulator.cc:465-471, same operation order, binary32
    volatile float k = p.gain[FAM_WBFM] / 75000;
    k = k * 32767;
    p.wbfm_k = k;
    volatile float f = p.gain[FAM_FM] / 15000;
    f = f * 32767;
    p.fm_k = f;
}

bool squelch_always_open(const ChanParams &p, const Consts &c)
{
    // SignalDetector.cc:259-266: present iff table(avg) - 42 - gain >= threshold, avg clipped to 127
    for (uint32_t m = 0; m <= 127; m++) {
        int32_t dbfs = c.db_table[m] - 42;
        dbfs = (int32_t)((uint32_t)dbfs - p.rx_gain_db);
        if (dbfs < p.threshold) return false;
    }
    return true;
}

TilePlan plan_tiles(uint32_t vlen, uint32_t n_channels, uint32_t chunk, uint32_t halo, uint32_t resident_wgs, uint32_t force_chunks)
{
    // One workgroup per tile, a tile = k whole chunks.  The launch runs in "rounds" of resident_wgs workgroups;
    // each round costs the tile plus its lead-in, and a partly filled last round costs as much as a full one.
    // Pick the k with the least total; ties go to the longer tile (less lead-in work overall).
    TilePlan p;
    p.tile_len = chunk;
    p.tiles_per_ch = 1;
    if (vlen == 0 || n_channels == 0) return p;
    if (resident_wgs == 0) resident_wgs = 768;
    const uint32_t overhead = chunk / 4;   // set-up, pipeline fill and drain of a tile, in samples
    uint64_t best = ~0ull;
    for (uint32_t k = 1; k <= 24; k++) {
        const uint64_t len = (uint64_t)k * chunk;
        const uint64_t per_ch = (vlen + len - 1) / len;
        const uint64_t rounds = (per_ch * n_channels + resident_wgs - 1) / resident_wgs;
        const uint64_t longest = len < vlen ? len : vlen;
        const uint64_t cost = rounds * (longest + halo + overhead);
        if (cost <= best) {
            best = cost;
            p.tile_len = (uint32_t)len;
            p.tiles_per_ch = (uint32_t)per_ch;
        }
        if (len >= vlen) break;
    }
    if (force_chunks) {   // experiments (IQD_PLAN_CHUNKS, read once by iqd_create): k chunks per tile
        p.tile_len = force_chunks * chunk;
        p.tiles_per_ch = (vlen + p.tile_len - 1) / p.tile_len;
    }
    return p;
}

// ---- streaming WBFM kernel (iqd_stream.hip) -----------------------------------------------------------------
// Byte j of lane group g of an MFMA operand is raw byte 16 g + j of a 64-byte run of interleaved I/Q: sample
// 8 g + j / 2, component j & 1.  Output row rho of the "N" window is sample 16 + rho of that run; in the "S"
// window the run is [piece bytes 0..31 | previous piece's bytes 32..63] and row rho is sample rho of the piece.
void build_stream_amat(int rotation, const int16_t *pre_q15, uint32_t *out)
{
    for (int type = 0; type < 2; type++)
        for (int rail = 0; rail < 2; rail++)
            for (int plane = 0; plane < 2; plane++) {
                uint32_t *m = out + (size_t)(4 * type + 2 * rail + plane) * 64 * 4;
                for (int lane = 0; lane < 64; lane++) {
                    const int rho = lane & 15, g = lane >> 4;
                    for (int j = 0; j < 16; j++) {
                        int i_rel = 8 * g + (j >> 1);
                        if (type == 1 && g >= 2) i_rel -= 32;
                        const int outpos = type == 0 ? 16 + rho : rho;
                        const int k = outpos - i_rel;                     // y[n] = sum h[k] x[n-k]
                        const int phase = ((i_rel % 4) + 4) % 4, comp = j & 1;
                        const bool feeds_i = rotation == 0 ? comp == 0 : comp == (phase & 1);
                        int8_t v = 0;
                        if (k >= 0 && k < 16 && feeds_i == (rail == 0)) {
                            const int t2 = 2 * (int)pre_q15[k];           // doubled: the index byte lands on bits 16..23
                            const int8_t lo = (int8_t)(t2 & 0xff);
                            const int hi = (t2 - lo) / 256;               // |2 h| <= 31866 -> |hi| <= 125
                            v = plane == 0 ? lo : (int8_t)hi;
                        }
                        m[lane * 4 + (j >> 2)] &= ~(0xffu << (8 * (j & 3)));
                        m[lane * 4 + (j >> 2)] |= (uint32_t)(uint8_t)v << (8 * (j & 3));
                    }
                }
            }
}

// A operands of the /4 first stages of the FM (32 taps) and AM/SSB (8 taps) chains (iqd_stream2.hip).  Output row
// 4 g' + r of the MFMA is output m = 2 g' + (r >> 1) of the 32-sample piece, rail r & 1 (0 = I', 1 = Q'); byte j of
// lane group g of the data operand is sample 8 g + j / 2, component j & 1, of this piece ([0], [1]) or of the
// previous one ([2], [3]).
void build_decim4_amat(int rotation, const int16_t *taps_q15, int ntaps, uint32_t *out)
{
    for (int which = 0; which < 2; which++)
        for (int plane = 0; plane < 2; plane++) {
            uint32_t *m = out + (size_t)(2 * which + plane) * 64 * 4;
            for (int lane = 0; lane < 64; lane++) {
                const int rho = lane & 15, g = lane >> 4;
                const int mo = 2 * (rho >> 2) + ((rho & 3) >> 1), rail = rho & 1;
                for (int j = 0; j < 16; j++) {
                    const int i_rel = 8 * g + (j >> 1) - (which ? 32 : 0);
                    const int k = 4 * mo + 3 - i_rel;                 // y[m] = sum h[k] x[4m+3-k]
                    const int phase = ((i_rel % 4) + 4) % 4, comp = j & 1;
                    const bool feeds_i = rotation == 0 ? comp == 0 : comp == (phase & 1);
                    int8_t v = 0;
                    if (k >= 0 && k < ntaps && feeds_i == (rail == 0)) {
                        const int tap = taps_q15[k];
                        const int8_t lo = (int8_t)(tap & 0xff);
                        v = plane == 0 ? lo : (int8_t)((tap - lo) / 256);
                    }
                    m[lane * 4 + (j >> 2)] &= ~(0xffu << (8 * (j & 3)));
                    m[lane * 4 + (j >> 2)] |= (uint32_t)(uint8_t)v << (8 * (j & 3));
                }
            }
        }
}

void build_d4_taps(const Consts &c, D4Args &da)
{
    auto pair = [](int16_t lo, int16_t hi) { return (uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16); };
    // stage 2 with DOUBLED taps (|2 h| <= 11828): the output (acc >> 15) then is the accumulator's high half and two
    // of them pack with one v_perm_b32; its input is a stage-1 output of int8 data (|y1| <= 114): nowhere near 2^31
    for (int q = 0; q < 6; q++) da.s2p[q] = pair((int16_t)(2 * c.am_s2[2 * q + 1]), (int16_t)(2 * c.am_s2[2 * q]));
    for (int q = 0; q < 8; q++) da.s3p[q] = pair(c.am_s3[2 * q + 1], c.am_s3[2 * q]);
    for (int j = 0; j < 16; j++) da.hilb[j] = (uint32_t)(uint16_t)c.ssb_hilbert[2 * j];
    for (int q = 0; q < 6; q++) da.p12p[q] = pair(c.post12[2 * q + 1], c.post12[2 * q]);
    for (int q = 0; q < 20; q++) da.a40p[q] = pair(c.audio40[2 * q + 1], c.audio40[2 * q]);
}

// |atan2(-r, x - 128)| for r = 0..128.  The kernel negates every angle (theta' = -theta, with -K), so that the
// sign bit of theta' is simply bit 7 of the y index: y < 0 -> theta < 0 -> theta' = +|theta|; y >= 0 ->
// theta' = -|theta| (y = 0: -0 or -pi).  Valid because the reference's table is odd in y bit for bit; checked here.
bool build_half_lut(float *out)
{
    for (int r = 0; r <= 128; r++)
        for (int x = 0; x < 256; x++) {
            const float neg = (float)atan2((double)-r, (double)x - 128);
            out[r * ST_ROW_FLOATS + x] = fabsf(neg);
            if (r >= 1 && r <= 127) {
                const float pos = (float)atan2((double)r, (double)x - 128);
                if (f2u(pos) != (f2u(neg) ^ 0x80000000u)) return false;
            }
            if (r == 0 && (f2u(neg) & 0x80000000u)) return false;     // atan2(-0.0 ...) is not what (double)0 gives
        }
    for (int r = 0; r <= 128; r++)
        for (int x = 256; x < ST_ROW_FLOATS; x++) out[r * ST_ROW_FLOATS + x] = 0.f;
    return true;
}

void build_stream_taps(const int16_t *d1, const int16_t *post12, const int16_t *audio40, StreamArgs &sa)
{
    auto pair = [](int16_t lo, int16_t hi) { return (uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16); };
    // stage 1: window x[4m-4 .. 4m+3] ascending <-> taps h[7 .. 0]
    for (int q = 0; q < 4; q++) sa.d1p[q] = pair(d1[7 - 2 * q], d1[6 - 2 * q]);
    for (int q = 0; q < 4; q++) sa.d1p2[q] = pair((int16_t)(2 * d1[7 - 2 * q]), (int16_t)(2 * d1[6 - 2 * q]));   // |2 h| <= 12892
    // stages 2 and 3: pair q counts back from the newest dword (older sample low: h[2q+1], newer high: h[2q])
    for (int q = 0; q < 6; q++) sa.p12p[q] = pair(post12[2 * q + 1], post12[2 * q]);
    for (int q = 0; q < 20; q++) sa.a40p[q] = pair(audio40[2 * q + 1], audio40[2 * q]);
}

// Segments of the streaming kernel: about `streams` of them in all, whole 128-sample units.
// A CU's LDS holds one persistent workgroup, so families that stream side by side have to share the CUs.
//  - whole multiples of 8: workgroups are dealt round-robin to the 8 XCDs, and the shares must fit side by side on every
//    one of them - or a family's last workgroups wait for a whole kernel of another family (seen: 38 + 59 + 91 + 65
//    workgroups put 34 on one XCD of 32 CUs, AM took twice as long);
//  - two CUs per XCD stay unplanned: with every CU spoken for, a workgroup that finds its CU still busy for a moment
//    waits for a whole kernel (0.35 ms per step of the mixed bench configuration with 240 of 256 CUs planned, 0.42-0.45
//    with all 256);
//  - what the rounding leaves over goes, 8 at a time, to whoever is furthest below its due.
bool plan_family_shares(const float *cost, int n, uint32_t n_cus, uint32_t *share)
{
    for (int f = 0; f < n; f++) share[f] = n_cus;
    float total = 0.f;
    for (int f = 0; f < n; f++) total += cost[f] > 0.f ? cost[f] : 0.f;
    if (n_cus < 64 || n > 8 || total <= 0.f) return false;
    const uint32_t budget = n_cus - 16;
    float want[8];
    uint32_t given = 0;
    for (int f = 0; f < n; f++) {
        if (!(cost[f] > 0.f)) { want[f] = 0.f; continue; }
        want[f] = (float)budget * cost[f] / total;
        uint32_t w = (uint32_t)want[f] & ~7u;
        if (w < 8) w = 8;
        share[f] = w;
        given += w;
    }
    while (given + 8 <= budget) {
        int best = -1;
        for (int f = 0; f < n; f++)
            if (cost[f] > 0.f && (best < 0 || want[f] - (float)share[f] > want[best] - (float)share[best])) best = f;
        if (best < 0) break;
        share[best] += 8;
        given += 8;
    }
    if (given > budget) {                                    // (many tiny families: no plan)
        for (int f = 0; f < n; f++) share[f] = n_cus;
        return false;
    }
    return true;
}

// The same for ONE launch whose workgroups are split among the families (iqd_stream_mixed.hip): every workgroup has a CU
// to itself and consecutive workgroups go round the XCDs, so the shares are plain integers in proportion to the cost
// (largest remainders first), at least one workgroup for a family that is there, all n_wgs given out.
void plan_fused_shares(const float *cost, int n, uint32_t n_wgs, uint32_t *share)
{
    float total = 0.f;
    int present = 0;
    for (int f = 0; f < n; f++) {
        share[f] = 0;
        if (cost[f] > 0.f) { total += cost[f]; present++; }
    }
    if (!present || n_wgs < (uint32_t)present) return;
    uint32_t given = 0;
    float rest[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int f = 0; f < n && f < 8; f++) {
        if (!(cost[f] > 0.f)) continue;
        const float want = (float)n_wgs * cost[f] / total;
        uint32_t w = (uint32_t)want;
        if (w < 1) w = 1;
        share[f] = w;
        rest[f] = want - (float)w;
        given += w;
    }
    while (given < n_wgs) {       // hand out what the rounding left, largest remainder first
        int best = -1;
        for (int f = 0; f < n && f < 8; f++)
            if (cost[f] > 0.f && (best < 0 || rest[f] > rest[best])) best = f;
        share[best]++;
        rest[best] -= 1.f;
        given++;
    }
    while (given > n_wgs) {       // (the minimum of one per family overdrew it: take from the largest)
        int big = 0;
        for (int f = 1; f < n && f < 8; f++)
            if (share[f] > share[big]) big = f;
        share[big]--;
        given--;
    }
}

// Shares by TIME (round 4).  Proportional shares ignore what a family's segments look like once they are cut: a segment is a
// whole number of granules behind a fixed lead-in, and a channel is a whole number of segments, so with many short rows a
// family sits between "one segment per channel" and "two" - 16 384 channels x 2^14 gave AM and SSB one 16 384-sample segment per
// channel on half-empty shares while WBFM and FM finished in two thirds of the time.  Here every family lists what each
// workgroup count would cost it - time = ns_per_sample x (segment + lead-in) for the shortest segment that fits one round of
// that many workgroups - and the smallest common deadline T is taken for which the families' cheapest workgroup counts fit
// the launch.  rot_count: the family's channels per rotation selector (segment ids are padded to 16 per selector group); pass
// them all in [0] for a family whose launch is not grouped.  Returns false when no single-round plan exists (the caller keeps
// the proportional shares, which run several rounds).
bool plan_fused_by_time(uint32_t vlen, int n, const FusedFamily *fam, uint32_t n_wgs, uint32_t *share)
{
    struct Opt { uint32_t wgs; float t; };
    std::vector<Opt> opts[8];
    std::vector<float> deadlines;
    if (n > 8) return false;
    for (int f = 0; f < n; f++) {
        share[f] = 0;
        const uint32_t n_ch = fam[f].rot_count[0] + fam[f].rot_count[1] + fam[f].rot_count[2];
        if (!n_ch) continue;
        const uint32_t g = fam[f].granule == 128 || fam[f].granule == 256 ? fam[f].granule : 512;
        uint32_t last_len = 0;
        // (short lead-ins: a channel's cold segments spend `shift` of their length on their lead-in, plan_stream; a channel of k
        //  segments has at most d4_max_cold(k) of them - here the bound for up to 65 segments, two)
        const uint64_t span = (uint64_t)vlen + (fam[f].shift ? 2ull * fam[f].shift : 0ull);
        const uint32_t min_tile = fam[f].shift ? std::max<uint32_t>((uint32_t)ST_MIN_TILE, fam[f].shift + 128) : (uint32_t)ST_MIN_TILE;
        for (uint32_t k = 1; last_len != min_tile && k <= vlen;) {   // (down to the shortest segment there is)
            uint64_t len = (span + k - 1) / k;
            len = (len + g - 1) / g * g;
            if (len < min_tile) len = min_tile;
            // the next k worth a look is the first that gives a shorter segment: ceil(vlen / k) <= len - g (one k at a
            // time this walked vlen / 768 values of k - 2 ms of host time for a 2^28-sample row, more than its kernels run)
            const uint64_t shorter = len > g ? len - g : 1;
            const uint32_t k_next = (uint32_t)std::max<uint64_t>(k + 1, (span + shorter - 1) / shorter);
            if ((uint32_t)len == last_len) { k = k_next; continue; }
            last_len = (uint32_t)len;
            k = k_next;
            uint32_t tiles = (uint32_t)(((uint64_t)vlen + len - 1) / len);
            if (!tiles) tiles = 1;
            while ((uint64_t)tiles * len < (uint64_t)vlen + (fam[f].shift ? (uint64_t)fam[f].shift * d4_max_cold(tiles) : 0u)) tiles++;   // (as plan_stream)
            uint64_t ids = 0;
            for (int r = 0; r < 3; r++) ids += ((uint64_t)fam[f].rot_count[r] * tiles + 15) / 16 * 16;
            const uint64_t wgs = (ids + ST_SEGS - 1) / ST_SEGS;
            if (wgs > n_wgs) break;                       // (shorter segments need even more)
            const float t = fam[f].ns_per_sample * (float)(len + fam[f].halo);
            if (!opts[f].empty() && opts[f].back().wgs == (uint32_t)wgs) opts[f].back().t = t;   // same workgroups, shorter segments
            else opts[f].push_back(Opt{(uint32_t)wgs, t});
            deadlines.push_back(t);
        }
        if (opts[f].empty()) return false;                // even one segment per channel does not fit one round
    }
    std::sort(deadlines.begin(), deadlines.end());
    for (float T : deadlines) {
        uint64_t total = 0;
        bool ok = true;
        for (int f = 0; f < n && ok; f++) {
            if (opts[f].empty()) continue;
            uint32_t best = 0;
            for (const Opt &o : opts[f])                  // (ascending workgroups, descending time: the first that meets T)
                if (o.t <= T) { best = o.wgs; break; }
            if (!best) ok = false;
            share[f] = best;
            total += best;
        }
        if (ok && total <= n_wgs) {
            // what the deadline leaves over goes round in proportion: more workgroups than a family needs mean shorter segments
            // for it, an earlier end and less company for the family that sets the pace (small calls: 1 % on the clock)
            uint32_t left = n_wgs - (uint32_t)total, given = 0;
            uint32_t extra[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int f = 0; f < n; f++) {
                extra[f] = (uint32_t)((uint64_t)left * share[f] / total);
                given += extra[f];
            }
            for (int f = 0; given < left; f = (f + 1) % n)
                if (share[f]) { extra[f]++; given++; }
            for (int f = 0; f < n; f++) share[f] += extra[f];
            return true;
        }
    }
    for (int f = 0; f < n; f++) share[f] = 0;
    return false;
}

TilePlan plan_stream(uint32_t vlen, uint32_t n_channels, uint32_t streams, uint32_t granule, uint32_t shift)
{
    TilePlan p;
    // As many segments as fit ONE round of the persistent workgroups (all segments of a round advance in lock step,
    // so a second round with a handful of stragglers would cost as much as the first).
    // shift (FM / AM / SSB with short lead-ins, iqd_stream.h: d4_geom): a channel's cold segments - its first one and those that
    // are lane 0 of a consumer wave, at most d4_max_cold(n) of its n segments - spend `shift` samples of their length on their
    // lead-in, so n segments of length L cover n * L - shift * cold samples; a segment is longer than the shift.
    uint32_t per_ch = n_channels ? streams / n_channels : 1;
    if (per_ch == 0) per_ch = 1;
    if (granule != 128 && granule != 256) granule = 512;
    const uint64_t min_tile = shift ? std::max<uint64_t>(ST_MIN_TILE, shift + 128) : (uint64_t)ST_MIN_TILE;
    auto span_of = [&](uint64_t n_tiles) { return (uint64_t)vlen + (shift ? (uint64_t)shift * d4_max_cold((uint32_t)n_tiles) : 0u); };
    uint64_t len = (span_of(per_ch) + per_ch - 1) / per_ch;
    len = (len + granule - 1) / granule * granule;   // 512: whole 32-byte PCM sectors per segment (16 PCM samples)
    if (len < min_tile) len = min_tile;   // a segment's end histories must be its own
    // (the minimum itself is not a multiple of the 512 granule: take it where it fits the round - 4096-sample rows on 6 segments
    //  per channel are 6 x 768, not 4 x 1024)
    if (len > min_tile && (span_of(per_ch) + min_tile - 1) / min_tile <= per_ch) len = min_tile;
    p.tile_len = (uint32_t)len;
    // the fewest segments that cover the row whatever the channel's first segment id
    uint64_t n = ((uint64_t)vlen + len - 1) / len;
    if (n == 0) n = 1;
    while (n * len < span_of(n)) n++;
    p.tiles_per_ch = (uint32_t)n;
    return p;
}

uint32_t block_magic(uint32_t block_samples)
{
    return (uint32_t)(((1ull << 32) + block_samples - 1) / block_samples);
}

}  // namespace iqd
