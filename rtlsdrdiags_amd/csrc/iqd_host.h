// Host-side logic of the engine that needs no GPU: constant tables, derived per-channel
// parameters, squelch pre-analysis and tile planning.  (Unit-tested on the CPU.)
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "iqd_device.h"

namespace iqd {

// (int16_t)float with the x86-64/gcc semantics the reference was built with.
int16_t host_cast_i16(float f);
// Decimator_int16.cc:55-63: hq = (int16_t) round(h * 32768)
void quantize_q15(const float *h, int n, int16_t *hq);
void build_consts(Consts &c);
// WbFmDemodulator.cc:159-170: lut[y*256+x] = (float)atan2(y-128., x-128.)
void build_atan2_lut(std::vector<float> &lut);
// FmDemodulator.cc:476 for every (q, i) in [-FM_LUT_R, FM_LUT_R]^2
void build_fm_lut(std::vector<float> &lut);

void default_params(ChanParams &p);
void derive_params(ChanParams &p);   // wbfm_k / fm_k from the gains, in binary32

// True when SignalDetector can never report "absent" for this channel, whatever the data
// (then Squelch::run() allows every block and no gating pass is needed).
bool squelch_always_open(const ChanParams &p, const Consts &c);

struct TilePlan { uint32_t tile_len, tiles_per_ch; };
TilePlan plan_tiles(uint32_t vlen, uint32_t n_channels, uint32_t chunk, uint32_t halo, uint32_t resident_wgs,
                    uint32_t force_chunks = 0 /* experiments: k chunks per tile */);

TilePlan plan_stream(uint32_t vlen, uint32_t n_channels, uint32_t streams, uint32_t granule = 512 /* segment lengths are multiples of this: 128, 256 or 512 samples */,
                     uint32_t shift = 0 /* FM / AM / SSB with short lead-ins (iqd_stream.h: d4_geom): the full lead-in minus 128 that a cold segment takes out of its own length */);
// Several demodulator families in one call, each with its streaming kernel: the CUs each family's persistent workgroups
// get, in proportion to cost[f] (0 = family absent: its entry becomes n_cus).  Whole multiples of 8, at least 8, two CUs
// per XCD left unplanned; false (and every entry n_cus) when that cannot be had.  See iqd_host.cpp.
bool plan_family_shares(const float *cost, int n, uint32_t n_cus, uint32_t *share);
void plan_fused_shares(const float *cost, int n, uint32_t n_wgs, uint32_t *share);   // several families as ranges of one launch
struct FusedFamily { uint32_t rot_count[3]; uint32_t halo, granule; float ns_per_sample; uint32_t shift = 0; /* see plan_stream */ };
bool plan_fused_by_time(uint32_t vlen, int n, const FusedFamily *fam, uint32_t n_wgs, uint32_t *share);

uint32_t block_magic(uint32_t block_samples);

}  // namespace iqd
