// oracle/ref_shim.cc — TEST INFRASTRUCTURE ONLY.
//
// A small C-ABI harness around the *unmodified* RtlSdrDiags reference classes.
// It is compiled together with the reference sources where they lie under
// /root/reference (see oracle/Makefile, target _ref/libiqd_ref.so); nothing
// from the reference is copied into this repository.  The resulting library
// is used only by tests/, by the golden-vector generator and, optionally, as
// the "reference" CPU baseline in bench.py.  The product path never loads it.
//
// The two symbols defined at the top are the only externals the hot-path
// sources need from the (un-buildable here) application files:
//   radio_adjustableReceiveGainInDb  — defined in src_diags/Radio.cc:19
//   nprintf                          — defined in src_diags/diagUi.cc:2166
// They are part of this harness (like demodulatorResearch/demodulators/demod.cc
// is a harness), not stand-ins for a missing library: Radio.cc / diagUi.cc need
// librtlsdr + libusb and are outside the hot path.
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>

#include "IqDataProcessor.h"
#include "Decimator_int16.h"
#include "FirFilter_int16.h"
#include "FirFilter.h"
#include "IirFilter.h"
#include "Squelch.h"
#include "DbfsCalculator.h"

uint32_t radio_adjustableReceiveGainInDb = 24;  // default: Radio.cc:325-328

void nprintf(FILE *s, const char *formatPtr, ...)
{
  va_list args;
  va_start(args, formatPtr);
  vfprintf(s, formatPtr, args);
  va_end(args);
}

namespace {

// The reference's PCM callback has no context pointer (radioApp.cc:103-111),
// so the harness routes it through one "current sink".
struct PcmSink
{
  int16_t *dst;
  size_t capacity;
  size_t count;
};

PcmSink *g_sink = 0;

void pcmCallback(int16_t *bufferPtr, uint32_t bufferLength)
{
  if (g_sink == 0) return;
  for (uint32_t i = 0; i < bufferLength; i++)
  {
    if (g_sink->count < g_sink->capacity)
      g_sink->dst[g_sink->count] = bufferPtr[i];
    g_sink->count++;
  }
}

struct RefChain
{
  IqDataProcessor *proc;
  AmDemodulator *am;
  FmDemodulator *fm;
  WbFmDemodulator *wbfm;
  SsbDemodulator *ssb;
  int lastAllowed;
  uint32_t lastMagnitude;
};

void signalStateCb(bool present, void *ctx)
{
  ((RefChain *)ctx)->lastAllowed = present ? 1 : 0;
}

void signalMagnitudeCb(uint32_t magnitude, void *ctx)
{
  ((RefChain *)ctx)->lastMagnitude = magnitude;
}

}  // namespace

extern "C" {

void *ref_create(void)
{
  static char host[] = "127.0.0.1";
  RefChain *c = new RefChain;
  c->proc = new IqDataProcessor(host, 8001);
  c->am = new AmDemodulator(pcmCallback);
  c->fm = new FmDemodulator(pcmCallback);
  c->wbfm = new WbFmDemodulator(pcmCallback);
  c->ssb = new SsbDemodulator(pcmCallback);
  c->proc->setAmDemodulator(c->am);
  c->proc->setFmDemodulator(c->fm);
  c->proc->setWbFmDemodulator(c->wbfm);
  c->proc->setSsbDemodulator(c->ssb);
  c->proc->registerSignalStateCallback(signalStateCb, c);
  c->proc->enableSignalNotification();
  c->proc->registerSignalMagnitudeCallback(signalMagnitudeCb, c);
  c->proc->enableSignalMagnitudeNotification();
  c->lastAllowed = 0;
  c->lastMagnitude = 0;
  return c;
}

void ref_destroy(void *h)
{
  RefChain *c = (RefChain *)h;
  delete c->proc;
  delete c->am;
  delete c->fm;
  delete c->wbfm;
  delete c->ssb;
  delete c;
}

void ref_set_mode(void *h, int mode)
{
  ((RefChain *)h)->proc->setDemodulatorMode((IqDataProcessor::demodulatorType)mode);
}

// which: 1=AM 2=FM 3=WBFM 4=SSB
void ref_set_gain(void *h, int which, float gain)
{
  RefChain *c = (RefChain *)h;
  switch (which)
  {
    case 1: c->am->setDemodulatorGain(gain); break;
    case 2: c->fm->setDemodulatorGain(gain); break;
    case 3: c->wbfm->setDemodulatorGain(gain); break;
    case 4: c->ssb->setDemodulatorGain(gain); break;
  }
}

void ref_set_squelch(void *h, int32_t threshold)
{
  ((RefChain *)h)->proc->setSignalDetectThreshold(threshold);
}

void ref_set_rx_gain_db(uint32_t gainInDb)
{
  radio_adjustableReceiveGainInDb = gainInDb;
}

void ref_reset(void *h)
{
  RefChain *c = (RefChain *)h;
  c->am->resetDemodulator();
  c->fm->resetDemodulator();
  c->wbfm->resetDemodulator();
  c->ssb->resetDemodulator();
}

// One call of IqDataProcessor::acceptIqData on a private copy of `iq`
// (the reference mutates its buffer in place).  byteCount <= 32768.
// Returns the number of PCM samples the callback delivered.
long ref_accept(void *h, const uint8_t *iq, size_t byteCount,
                int16_t *pcm, size_t pcmCapacity,
                uint32_t *magnitude, uint8_t *allowed)
{
  static unsigned char scratch[32768];
  RefChain *c = (RefChain *)h;
  if (byteCount > sizeof(scratch)) return -1;
  memcpy(scratch, iq, byteCount);
  PcmSink sink = {pcm, pcmCapacity, 0};
  g_sink = &sink;
  c->proc->acceptIqData(0, scratch, byteCount);
  g_sink = 0;
  if (magnitude) *magnitude = c->lastMagnitude;
  if (allowed) *allowed = (uint8_t)c->lastAllowed;
  return (long)sink.count;
}

// Stream helper: feeds `total` bytes in blocks of `blockBytes`, appending PCM.
// magnitude/allowed (optional) receive one entry per block.
long ref_accept_stream(void *h, const uint8_t *iq, size_t total, size_t blockBytes,
                       int16_t *pcm, size_t pcmCapacity,
                       uint32_t *magnitude, uint8_t *allowed)
{
  size_t produced = 0;
  size_t block = 0;
  for (size_t off = 0; off < total; off += blockBytes, block++)
  {
    size_t n = (total - off < blockBytes) ? (total - off) : blockBytes;
    long got = ref_accept(h, iq + off, n, pcm + produced,
                          pcmCapacity > produced ? pcmCapacity - produced : 0,
                          magnitude ? magnitude + block : 0,
                          allowed ? allowed + block : 0);
    if (got < 0) return got;
    produced += (size_t)got;
  }
  return (long)produced;
}

// Demodulator-level entry (signed int8 IQ, no front end), like demod.cc.
// mode: 1=AM 2=FM 3=WBFM 4=LSB 5=USB.  The buffer is mutated (WBFM).
long ref_demod_accept(void *h, int mode, int8_t *iq, uint32_t byteCount,
                      int16_t *pcm, size_t pcmCapacity)
{
  RefChain *c = (RefChain *)h;
  PcmSink sink = {pcm, pcmCapacity, 0};
  g_sink = &sink;
  switch (mode)
  {
    case 1: c->am->acceptIqData(iq, byteCount); break;
    case 2: c->fm->acceptIqData(iq, byteCount); break;
    case 3: c->wbfm->acceptIqData(iq, byteCount); break;
    case 4: c->ssb->setLsbDemodulationMode(); c->ssb->acceptIqData(iq, byteCount); break;
    case 5: c->ssb->setUsbDemodulationMode(); c->ssb->acceptIqData(iq, byteCount); break;
  }
  g_sink = 0;
  return (long)sink.count;
}

// In-place +Fs/4 / -Fs/4 rotations (IqDataProcessor.cc:567-611 / :496-540).
void ref_upconvert(void *h, int8_t *buf, uint32_t byteCount)
{
  ((RefChain *)h)->proc->upconvertByFsOver4(buf, byteCount);
}

void ref_downconvert(void *h, int8_t *buf, uint32_t byteCount)
{
  ((RefChain *)h)->proc->downconvertByFsOver4(buf, byteCount);
}

// ---- filter-level known-answer entry points --------------------------------

// Decimator_int16 (Filters/Int16/Decimator_int16.cc): returns output count.
long ref_decimator_int16(const float *taps, int length, int factor,
                         const int16_t *in, size_t n, int16_t *out)
{
  Decimator_int16 d(length, (float *)taps, factor);
  size_t m = 0;
  for (size_t i = 0; i < n; i++)
  {
    int16_t y;
    if (d.decimate(in[i], &y)) out[m++] = y;
  }
  return (long)m;
}

// FirFilter_int16 (Filters/Int16/FirFilter_int16.cc).
void ref_fir_int16(const float *taps, int length,
                   const int16_t *in, size_t n, int16_t *out)
{
  FirFilter_int16 f(length, (float *)taps);
  for (size_t i = 0; i < n; i++) out[i] = f.filterData(in[i]);
}

// FirFilter (float, Filters/FirFilter.cc).
void ref_fir_f32(const float *taps, int length,
                 const float *in, size_t n, float *out)
{
  FirFilter f(length, (float *)taps);
  for (size_t i = 0; i < n; i++) out[i] = f.filterData(in[i]);
}

// IirFilter (Filters/IirFilter.cc).
void ref_iir_f32(const float *num, int numLength, const float *den, int denLength,
                 const float *in, size_t n, float *out)
{
  IirFilter f(numLength, (float *)num, denLength, (float *)den);
  for (size_t i = 0; i < n; i++) out[i] = f.filterData(in[i]);
}

// Squelch (Squelch.cc / SignalDetector.cc / SignalTracker.cc) on signed data.
int ref_squelch_run(void *sq, uint32_t gainInDb, const int8_t *buf, uint32_t byteCount,
                    uint32_t *magnitude)
{
  Squelch *s = (Squelch *)sq;
  bool allowed = s->run(gainInDb, (int8_t *)buf, byteCount);
  if (magnitude) *magnitude = s->getSignalMagnitude();
  return allowed ? 1 : 0;
}

void *ref_squelch_create(int32_t threshold) { return new Squelch(threshold); }
void ref_squelch_destroy(void *sq) { delete (Squelch *)sq; }

// DbfsCalculator(7)::convertMagnitudeToDbFs (DbfsCalculator.cc:111-147).
int32_t ref_dbfs(uint32_t magnitude)
{
  static DbfsCalculator calc(7);
  return calc.convertMagnitudeToDbFs(magnitude);
}

}  // extern "C"
