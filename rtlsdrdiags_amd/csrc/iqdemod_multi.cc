// iqdemod_multi — many channels from files through ONE engine, straight over the C ABI (include/iqdemod.h): the many-dongle
// counterpart of iqdemod_file (block source and sink adapters, multi-channel file layouts, host buffers
// that let the reading overlap the kernels).  Every channel is one reference IqDataProcessor with its demodulators
// (Radio.cc:150-181); a call hands each of them `blocks` consecutive 32768-byte blocks, which is `blocks` calls of
// IqDataProcessor::acceptIqData per channel (DataConsumer.cc:333-346), and the PCM goes out as S16_LE at 8 kS/s
// (radioApp.cc:103-111) - one file per channel.
//
//   iqdemod_multi channels=N in=<pattern> out=<pattern> [modes=<m>[,<m>...]] [layout=files|interleaved]
//                 [blocks=K] [threshold=<dBFS>] [rotation=<r>[,<r>...]] [agc=0|1]
//
//   layout=files        (default) N captures, `in` is a printf pattern with one %d (channel number): capture_%d.iq
//   layout=interleaved  ONE capture whose 32768-byte blocks go round the channels - block b of channel c is block
//                       b * N + c of the file (what a recorder of N dongles writes when it takes them in turn)
//   modes / rotation    per channel, the list repeating: modes=2,3 = FM, WBFM, FM, WBFM ... (0 none 1 am 2 fm 3 wbfm 4 lsb
//                       5 usb; IqDataProcessor.h:16-58), rotation +1 / 0 / -1 = the Fs/4 selector (+1 is the reference's)
//   out                 printf pattern with one %d: pcm_%d.s16
//
// Two page-locked input buffers (iqd_host_alloc): a reader thread fills one while the engine works on the other.  A
// capture that ends inside a call is processed up to its last whole 64-byte unit (a short block, Radio.cc:1895-1906);
// channels whose captures are shorter than the others simply stop earlier.  Exit status 0, 1 (no device / bad
// arguments / I/O), 3 (a call was rejected).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "iqdemod.h"

namespace {

const size_t BLOCK = 32768;

struct Options {
  uint32_t channels = 0, blocks = 16;
  std::string in, out;
  bool interleaved = false;
  std::vector<int> modes{2}, rotations{1};
  int threshold = -200, agc = -1;
};

std::vector<int> intList(const char *s)
{
  std::vector<int> v;
  for (const char *q = s; *q;) {
    v.push_back(atoi(q));
    const char *c = strchr(q, ',');
    if (!c) break;
    q = c + 1;
  }
  return v;
}

bool parse(int argc, char **argv, Options &o)
{
  for (int i = 1; i < argc; i++) {
    const char *a = argv[i];
    if (!strncmp(a, "channels=", 9)) o.channels = (uint32_t)atoi(a + 9);
    else if (!strncmp(a, "blocks=", 7)) o.blocks = (uint32_t)atoi(a + 7);
    else if (!strncmp(a, "in=", 3)) o.in = a + 3;
    else if (!strncmp(a, "out=", 4)) o.out = a + 4;
    else if (!strncmp(a, "modes=", 6)) o.modes = intList(a + 6);
    else if (!strncmp(a, "rotation=", 9)) o.rotations = intList(a + 9);
    else if (!strncmp(a, "threshold=", 10)) o.threshold = atoi(a + 10);
    else if (!strncmp(a, "agc=", 4)) o.agc = atoi(a + 4);
    else if (!strcmp(a, "layout=files")) o.interleaved = false;
    else if (!strcmp(a, "layout=interleaved")) o.interleaved = true;
    else return false;
  }
  return o.channels > 0 && o.blocks > 0 && !o.in.empty() && !o.out.empty() && !o.modes.empty() && !o.rotations.empty();
}

// One filled input buffer: [channels][blocks * BLOCK] bytes, of which channel c has got[c] (a multiple of 64)
struct Batch {
  uint8_t *iq = nullptr;
  std::vector<size_t> got;
  bool last = false;
};

struct Reader {
  const Options &o;
  std::vector<FILE *> files;
  explicit Reader(const Options &opt) : o(opt) {}
  bool open()
  {
    char name[4096];
    if (o.interleaved) {
      FILE *f = fopen(o.in.c_str(), "rb");
      if (!f) { perror(o.in.c_str()); return false; }
      files.push_back(f);
      return true;
    }
    for (uint32_t c = 0; c < o.channels; c++) {
      snprintf(name, sizeof name, o.in.c_str(), (int)c);
      FILE *f = fopen(name, "rb");
      if (!f) { perror(name); return false; }
      files.push_back(f);
    }
    return true;
  }
  // fills b; returns false when nothing at all was read
  bool fill(Batch &b)
  {
    const size_t row = (size_t)o.blocks * BLOCK;
    bool any = false;
    b.last = false;
    if (!o.interleaved) {
      for (uint32_t c = 0; c < o.channels; c++) {
        const size_t n = fread(b.iq + c * row, 1, row, files[c]);
        b.got[c] = n & ~(size_t)63;
        any = any || b.got[c] != 0;
      }
      return any;   // (captures of different lengths: the batches go on until no channel has anything left)
    }
    for (uint32_t c = 0; c < o.channels; c++) b.got[c] = 0;
    for (uint32_t k = 0; k < o.blocks && !b.last; k++)
      for (uint32_t c = 0; c < o.channels; c++) {
        const size_t n = fread(b.iq + c * row + (size_t)k * BLOCK, 1, BLOCK, files[0]);
        b.got[c] += n & ~(size_t)63;
        any = any || n != 0;
        if (n < BLOCK) { b.last = true; break; }
      }
    return any;
  }
  ~Reader()
  {
    for (FILE *f : files) fclose(f);
  }
};

}  // namespace

int main(int argc, char **argv)
{
  Options o;
  if (!parse(argc, argv, o)) {
    fprintf(stderr, "usage: %s channels=N in=<pattern %%d | file> out=<pattern %%d> [modes=2,3,...] [layout=files|interleaved] "
                    "[blocks=K] [threshold=dBFS] [rotation=1,0,-1,...] [agc=0|1]\n", argv[0]);
    return 1;
  }
  iqd_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.abi_version = IQD_ABI_VERSION;
  cfg.n_channels = o.channels;
  cfg.block_bytes = (uint32_t)BLOCK;
  cfg.device = -1;
  iqd_t *e = nullptr;
  int rc = iqd_create(&cfg, &e);
  if (rc != IQD_OK) {
    fprintf(stderr, "iqdemod_multi: %s\n", iqd_strerror(rc));   // IQD_ENODEV without a GPU: there is no CPU path
    return 1;
  }
  for (uint32_t c = 0; c < o.channels; c++) {
    iqd_set_mode(e, c, 1, o.modes[c % o.modes.size()]);
    iqd_set_rotation(e, c, 1, o.rotations[c % o.rotations.size()]);
  }
  iqd_set_squelch(e, 0, o.channels, o.threshold);
  if (o.agc >= 0) {
    iqd_agc_set_type(e, 0, o.channels, o.agc);
    iqd_agc_enable(e, 0, o.channels, 1);
  }

  Reader reader(o);
  if (!reader.open()) return 1;
  std::vector<FILE *> outs(o.channels);
  for (uint32_t c = 0; c < o.channels; c++) {
    char name[4096];
    snprintf(name, sizeof name, o.out.c_str(), (int)c);
    outs[c] = fopen(name, "wb");
    if (!outs[c]) { perror(name); return 1; }
  }

  const size_t row = (size_t)o.blocks * BLOCK, pcmRow = row / 64;
  Batch batch[2];
  for (Batch &b : batch) {
    void *p = nullptr;
    if (iqd_host_alloc(e, (size_t)o.channels * row, &p) != IQD_OK) { fprintf(stderr, "iqdemod_multi: %s\n", iqd_last_error(e)); return 1; }
    b.iq = (uint8_t *)p;
    b.got.assign(o.channels, 0);
  }
  void *pp = nullptr;
  if (iqd_host_alloc(e, (size_t)o.channels * pcmRow * sizeof(int16_t), &pp) != IQD_OK) return 1;
  int16_t *pcm = (int16_t *)pp;
  std::vector<uint32_t> count(o.channels);

  // the reader runs one batch ahead of the engine
  std::mutex mu;
  std::condition_variable cv;
  int filled[2] = {0, 0};   // 0 free, 1 ready, 2 ready and nothing behind it
  bool readerDone = false;
  std::thread rd([&] {
    for (int k = 0;; k ^= 1) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return filled[k] == 0; });
      }
      const bool any = reader.fill(batch[k]);
      std::unique_lock<std::mutex> lk(mu);
      filled[k] = any ? (batch[k].last ? 2 : 1) : 2;
      if (!any) for (size_t &g : batch[k].got) g = 0;
      cv.notify_all();
      if (filled[k] == 2) { readerDone = true; return; }
    }
  });

  int status = 0;
  uint64_t samples = 0, pcmOut = 0;
  for (int k = 0;; k ^= 1) {
    int state;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return filled[k] != 0; });
      state = filled[k];
    }
    Batch &b = batch[k];
    // channels that still have the whole batch go in one call; shorter remainders (the capture's end) one by one
    size_t full = 0;
    for (uint32_t c = 0; c < o.channels; c++) full += b.got[c] == row ? 1 : 0;
    auto emit = [&](uint32_t c0, uint32_t n, size_t bytes) {
      // the rows of this batch lie `row` bytes apart whatever their length: hand a short call over row by row
      for (uint32_t c = c0; c < c0 + n; c++) {
        const int r = iqd_accept_iq(e, c, 1, b.iq + (size_t)c * row, bytes, pcm + (size_t)c * pcmRow, &count[c], nullptr, nullptr);
        if (r != IQD_OK) { fprintf(stderr, "iqdemod_multi: channel %u: %s\n", c, iqd_last_error(e)); status = 3; count[c] = 0; }
      }
    };
    if (full == o.channels) {
      const int r = iqd_accept_iq(e, 0, o.channels, b.iq, row, pcm, count.data(), nullptr, nullptr);
      if (r != IQD_OK) { fprintf(stderr, "iqdemod_multi: %s\n", iqd_last_error(e)); status = 3; std::fill(count.begin(), count.end(), 0u); }
    } else {
      for (uint32_t c = 0; c < o.channels; c++) {
        count[c] = 0;
        // whole blocks first, then one short block (a call is k whole blocks or ONE short one: include/iqdemod.h)
        const size_t whole = b.got[c] / BLOCK * BLOCK, rest = b.got[c] - whole;
        uint32_t n0 = 0;
        if (whole) { emit(c, 1, whole); n0 = count[c]; }
        if (rest) {
          const int r = iqd_accept_iq(e, c, 1, b.iq + (size_t)c * row + whole, rest, pcm + (size_t)c * pcmRow + n0, &count[c], nullptr, nullptr);
          if (r != IQD_OK) { fprintf(stderr, "iqdemod_multi: channel %u: %s\n", c, iqd_last_error(e)); status = 3; count[c] = 0; }
          count[c] += n0;
        }
      }
    }
    for (uint32_t c = 0; c < o.channels; c++) {
      if (count[c]) fwrite(pcm + (size_t)c * pcmRow, sizeof(int16_t), count[c], outs[c]);
      samples += b.got[c] / 2;
      pcmOut += count[c];
    }
    {
      std::unique_lock<std::mutex> lk(mu);
      filled[k] = 0;
      cv.notify_all();
    }
    if (state == 2) break;
  }
  rd.join();
  (void)readerDone;
  for (FILE *f : outs) fclose(f);
  fprintf(stderr, "iqdemod_multi: %u channels, %llu IQ samples in, %llu PCM samples out\n", o.channels,
          (unsigned long long)samples, (unsigned long long)pcmOut);
  iqd_host_free(e, pcm);
  for (Batch &b : batch) iqd_host_free(e, b.iq);
  iqd_destroy(e);
  return status;
}
