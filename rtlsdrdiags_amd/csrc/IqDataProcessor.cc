// Host-side mirror of the reference's IqDataProcessor and demodulator classes over the C ABI.
#include "IqDataProcessor.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

// ---- demodulator handles ------------------------------------------------------------------
DemodulatorHandle::DemodulatorHandle(int demod, float defaultGain, PcmCallback cb)
    : demod(demod), demodulatorGain(defaultGain), pcmCallbackPtr(cb), engine(0), ownsEngine(false), lastStatus(IQD_OK)
{
}

DemodulatorHandle::~DemodulatorHandle(void)
{
  if (ownsEngine && engine != 0) iqd_destroy(engine);
}

// e.g. WbFmDemodulator.cc:383-411 + sendPcmData (:582-591): the demodulator alone, PCM to its callback
void DemodulatorHandle::acceptIqData(int8_t *bufferPtr, uint32_t bufferLength)
{
  if (engine == 0) {   // nobody's demodulator: an engine of its own (there is no CPU path: without a device every call fails)
    iqd_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.abi_version = IQD_ABI_VERSION;
    cfg.n_channels = 1;
    cfg.device = -1;
    lastStatus = iqd_create(&cfg, &engine);
    if (lastStatus != IQD_OK) {
      engine = 0;
      fprintf(stderr, "DemodulatorHandle::acceptIqData: %s\n", iqd_strerror(lastStatus));
      return;
    }
    ownsEngine = true;
    iqd_set_gain(engine, 0, 1, demod, demodulatorGain);
  }
  if (bufferLength > 32768) {   // demodulatedData[16384], e.g. WbFmDemodulator.h:52
    lastStatus = IQD_EINVAL;
    fprintf(stderr, "DemodulatorHandle::acceptIqData: %u bytes exceed the demodulator's 32768-byte block\n", bufferLength);
    return;
  }
  beforeAccept();
  lastStatus = iqd_demod_accept(engine, 0, 1, demod, bufferPtr, bufferLength, pcmData);
  if (lastStatus != IQD_OK) {
    fprintf(stderr, "DemodulatorHandle::acceptIqData: %s\n", iqd_last_error(engine));
    return;
  }
  if (pcmCallbackPtr != 0) pcmCallbackPtr(pcmData, bufferLength / 64);
}

void DemodulatorHandle::setDemodulatorGain(float gain)
{
  demodulatorGain = gain;
  if (engine != 0) iqd_set_gain(engine, 0, 1, demod, gain);
}

void DemodulatorHandle::resetDemodulator(void)
{
  if (engine != 0) iqd_reset_demod(engine, 0, 1, demod);
}

void DemodulatorHandle::displayInternalInformation(void)
{
  static const char *names[] = {"", "AM", "FM", "Wideband FM", "SSB"};
  fprintf(stderr, "\n%s Demodulator Internal Information\n", names[demod]);
  fprintf(stderr, "Demodulator Gain         : %f\n", demodulatorGain);
}

// default gains: AmDemodulator.cc:104, FmDemodulator.cc:158, WbFmDemodulator.cc:173, SsbDemodulator.cc:147
AmDemodulator::AmDemodulator(PcmCallback cb) : DemodulatorHandle(IQD_DEMOD_AM, 300, cb) {}
FmDemodulator::FmDemodulator(PcmCallback cb) : DemodulatorHandle(IQD_DEMOD_FM, 64000 / (2 * M_PI), cb) {}
WbFmDemodulator::WbFmDemodulator(PcmCallback cb) : DemodulatorHandle(IQD_DEMOD_WBFM, 256000 / (2 * M_PI), cb) {}
SsbDemodulator::SsbDemodulator(PcmCallback cb) : DemodulatorHandle(IQD_DEMOD_SSB, 300, cb), lsbDemodulationMode(true) {}

void SsbDemodulator::setLsbDemodulationMode(void) { lsbDemodulationMode = true; }
void SsbDemodulator::setUsbDemodulationMode(void) { lsbDemodulationMode = false; }
void SsbDemodulator::beforeAccept(void) { iqd_demod_set_sideband(engine, 0, 1, lsbDemodulationMode ? 1 : 0); }

// ---- IqDataProcessor ------------------------------------------------------------------------
IqDataProcessor::IqDataProcessor(char *hostIpAddress, int hostPort)
    : engine(0), demodulatorMode(None), signalDetectThreshold(-200), blockBytes(32768),
      amDemodulatorPtr(0), fmDemodulatorPtr(0), wbFmDemodulatorPtr(0), ssbDemodulatorPtr(0),
      signalNotificationEnabled(false), signalCallbackContextPtr(0), signalCallbackPtr(0),
      signalMagnitudeNotificationEnabled(false), signalMagnitudeCallbackContextPtr(0),
      signalMagnitudeCallbackPtr(0), iqDumpEnabled(false), iqDumpContextPtr(0), iqDumpCallbackPtr(0),
      lastStatus(IQD_OK), receiveBlockCount(0), rejectedBlocks(0)
{
  (void)hostIpAddress;
  (void)hostPort;
  iqd_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.abi_version = IQD_ABI_VERSION;
  cfg.n_channels = 1;
  cfg.block_bytes = blockBytes;
  cfg.device = -1;
  lastStatus = iqd_create(&cfg, &engine);
  if (lastStatus != IQD_OK) engine = 0;   // every accept then reports the failure; no CPU path
}

IqDataProcessor::~IqDataProcessor(void)
{
  DemodulatorHandle *all[4] = {amDemodulatorPtr, fmDemodulatorPtr, wbFmDemodulatorPtr, ssbDemodulatorPtr};
  for (int i = 0; i < 4; i++)
    if (all[i] != 0) all[i]->engine = 0;
  if (engine != 0) iqd_destroy(engine);
}

const char *IqDataProcessor::lastError(void) const
{
  return engine != 0 ? iqd_last_error(engine) : iqd_strerror(lastStatus);
}

void IqDataProcessor::attach(DemodulatorHandle *h)
{
  if (h == 0 || engine == 0) return;
  h->engine = engine;
  iqd_set_gain(engine, 0, 1, h->demod, h->demodulatorGain);
}

void IqDataProcessor::setAmDemodulator(AmDemodulator *p) { amDemodulatorPtr = p; attach(p); }
void IqDataProcessor::setFmDemodulator(FmDemodulator *p) { fmDemodulatorPtr = p; attach(p); }
void IqDataProcessor::setWbFmDemodulator(WbFmDemodulator *p) { wbFmDemodulatorPtr = p; attach(p); }
void IqDataProcessor::setSsbDemodulator(SsbDemodulator *p) { ssbDemodulatorPtr = p; attach(p); }

void IqDataProcessor::setDemodulatorMode(demodulatorType mode)
{
  demodulatorMode = mode;
  if (ssbDemodulatorPtr != 0) {   // IqDataProcessor.cc:244-256
    if (mode == Lsb) ssbDemodulatorPtr->setLsbDemodulationMode();
    if (mode == Usb) ssbDemodulatorPtr->setUsbDemodulationMode();
  }
  if (engine != 0) iqd_set_mode(engine, 0, 1, (int)mode);
}

void IqDataProcessor::setSignalDetectThreshold(int32_t threshold)
{
  signalDetectThreshold = threshold;
  if (engine != 0) iqd_set_squelch(engine, 0, 1, threshold);
}

void IqDataProcessor::setReceiveGainInDb(uint32_t gainInDb)
{
  if (engine != 0) iqd_set_rx_gain_db(engine, 0, 1, gainInDb);
}

void IqDataProcessor::enableSignalNotification(void) { signalNotificationEnabled = true; }
void IqDataProcessor::disableSignalNotification(void) { signalNotificationEnabled = false; }
void IqDataProcessor::registerSignalStateCallback(void (*cb)(bool, void *), void *contextPtr)
{
  signalCallbackContextPtr = contextPtr;
  signalCallbackPtr = cb;
}
void IqDataProcessor::enableSignalMagnitudeNotification(void) { signalMagnitudeNotificationEnabled = true; }
void IqDataProcessor::disableSignalMagnitudeNotification(void) { signalMagnitudeNotificationEnabled = false; }
void IqDataProcessor::registerSignalMagnitudeCallback(void (*cb)(uint32_t, void *), void *contextPtr)
{
  signalMagnitudeCallbackContextPtr = contextPtr;
  signalMagnitudeCallbackPtr = cb;
}

// IqDataProcessor.cc:722-840: squelch, the two notifications, then the selected demodulator,
// whose PCM goes to that demodulator's callback (e.g. WbFmDemodulator.cc:582-591).
void IqDataProcessor::acceptIqData(unsigned long timeStamp, unsigned char *bufferPtr, unsigned long byteCount)
{
  (void)timeStamp;
  if (engine == 0) {   // no HIP device: there is no CPU path, every block is reported
    rejectedBlocks++;
    return;
  }
  // Whatever the read returned is one block (short USB reads included; the squelch averages over this call).
  // What the engine cannot take - an empty call, more than the 32768 bytes of SignalDetector.h:49, or a
  // length that is not a whole number of 64-byte units - is refused by iqd_accept_iq and surfaced here.
  uint32_t pcmCount = 0, magnitude = 0;
  uint8_t allowed = 0;
  if (iqDumpEnabled && iqDumpCallbackPtr != 0 && byteCount <= blockBytes && byteCount % 8 == 0 &&   // IqDataProcessor.cc:756-760
      iqd_front_end(engine, 0, 1, bufferPtr, byteCount, dumpData) == IQD_OK)
    iqDumpCallbackPtr(dumpData, (uint32_t)byteCount, iqDumpContextPtr);
  lastStatus = byteCount > blockBytes ? IQD_EINVAL
                                      : iqd_accept_iq(engine, 0, 1, bufferPtr, byteCount, pcmData, &pcmCount, &magnitude, &allowed);
  if (lastStatus != IQD_OK) {
    rejectedBlocks++;
    fprintf(stderr, "IqDataProcessor::acceptIqData: %lu bytes not processed: %s\n", byteCount,
            byteCount > blockBytes ? "more than one 32768-byte block" : iqd_last_error(engine));
    return;
  }
  receiveBlockCount++;
  if (signalNotificationEnabled && signalCallbackPtr != 0)
    signalCallbackPtr(allowed != 0, signalCallbackContextPtr);
  if (signalMagnitudeNotificationEnabled && signalMagnitudeCallbackPtr != 0)
    signalMagnitudeCallbackPtr(magnitude, signalMagnitudeCallbackContextPtr);
  if (allowed == 0 || pcmCount == 0) return;
  DemodulatorHandle *h = 0;
  switch (demodulatorMode) {
    case Am: h = amDemodulatorPtr; break;
    case Fm: h = fmDemodulatorPtr; break;
    case WbFm: h = wbFmDemodulatorPtr; break;
    case Lsb: case Usb: h = ssbDemodulatorPtr; break;
    default: break;
  }
  if (h != 0 && h->pcmCallbackPtr != 0) h->pcmCallbackPtr(pcmData, pcmCount);
}

void IqDataProcessor::upconvertByFsOver4(int8_t *bufferPtr, uint32_t byteCount)
{
  if (engine != 0) lastStatus = iqd_convert_fs_over_4(engine, +1, bufferPtr, byteCount);
}

void IqDataProcessor::downconvertByFsOver4(int8_t *bufferPtr, uint32_t byteCount)
{
  if (engine != 0) lastStatus = iqd_convert_fs_over_4(engine, -1, bufferPtr, byteCount);
}

void IqDataProcessor::displayInternalInformation(void)
{
  static const char *modes[] = {"None", "AM", "FM", "WBFM", "LSB", "USB"};
  fprintf(stderr, "\nIq Data Processor Internal Information\n");
  fprintf(stderr, "Demodulator Mode         : %s\n", modes[(int)demodulatorMode]);
  fprintf(stderr, "Signal Detect Threshold  : %d dBFs\n", (int)signalDetectThreshold);
  fprintf(stderr, "Receive Block Count      : %lu\n", receiveBlockCount);
  fprintf(stderr, "Rejected Block Count     : %lu\n", rejectedBlocks);
  fprintf(stderr, "Engine                   : %s\n", engine != 0 ? "MI355X (HIP)" : "unavailable");
}

void IqDataProcessor::deviceOperationCounts(unsigned long long *launches, unsigned long long *copies) const
{
  iqd_stats st;
  memset(&st, 0, sizeof(st));
  if (engine != 0) (void)iqd_get_stats(engine, &st);
  if (launches != 0) *launches = st.device_launches;
  if (copies != 0) *copies = st.device_copies;
}

// ---- AutomaticGainControl ---------------------------------------------------------------------------
AutomaticGainControl::AutomaticGainControl(IqDataProcessor *processorPtr, int32_t operatingPointInDbFs)
{
  this->processorPtr = processorPtr;
  setOperatingPoint(operatingPointInDbFs);
}

AutomaticGainControl::~AutomaticGainControl(void) { disable(); }

void AutomaticGainControl::setOperatingPoint(int32_t operatingPointInDbFs)
{
  if (processorPtr->engine != 0) iqd_agc_set_operating_point(processorPtr->engine, 0, 1, operatingPointInDbFs);
}

bool AutomaticGainControl::setAgcFilterCoefficient(float coefficient)
{
  return processorPtr->engine != 0 && iqd_agc_set_filter_coefficient(processorPtr->engine, 0, 1, coefficient) == IQD_OK;
}

bool AutomaticGainControl::setType(uint32_t type)
{
  return processorPtr->engine != 0 && iqd_agc_set_type(processorPtr->engine, 0, 1, type) == IQD_OK;
}

bool AutomaticGainControl::setDeadband(uint32_t deadbandInDb)
{
  return processorPtr->engine != 0 && iqd_agc_set_deadband(processorPtr->engine, 0, 1, deadbandInDb) == IQD_OK;
}

bool AutomaticGainControl::setBlankingLimit(uint32_t blankingLimit)
{
  return processorPtr->engine != 0 && iqd_agc_set_blanking_limit(processorPtr->engine, 0, 1, blankingLimit) == IQD_OK;
}

bool AutomaticGainControl::enable(void)
{
  return processorPtr->engine != 0 && iqd_agc_enable(processorPtr->engine, 0, 1, 1) == IQD_OK;
}

bool AutomaticGainControl::disable(void)
{
  return processorPtr->engine != 0 && iqd_agc_enable(processorPtr->engine, 0, 1, 0) == IQD_OK;
}

bool AutomaticGainControl::isEnabled(void)
{
  iqd_agc_state st;
  return processorPtr->engine != 0 && iqd_agc_get_state(processorPtr->engine, 0, &st) == IQD_OK && st.enabled != 0;
}

uint32_t AutomaticGainControl::getSignalMagnitude(void)
{
  iqd_agc_state st;
  if (processorPtr->engine == 0 || iqd_agc_get_state(processorPtr->engine, 0, &st) != IQD_OK) return 0;
  return st.signal_magnitude;
}

uint32_t AutomaticGainControl::getReceiveIfGainInDb(void)
{
  uint32_t gain = 0;
  if (processorPtr->engine != 0) iqd_get_rx_gain_db(processorPtr->engine, 0, &gain);
  return gain;
}

// AutomaticGainControl.cc:1082-1149
void AutomaticGainControl::displayInternalInformation(void)
{
  iqd_agc_state st;
  if (processorPtr->engine == 0 || iqd_agc_get_state(processorPtr->engine, 0, &st) != IQD_OK) return;
  fprintf(stderr, "\n--------------------------------------------\n");
  fprintf(stderr, "AGC Internal Information\n");
  fprintf(stderr, "--------------------------------------------\n");
  fprintf(stderr, "AGC Enabled               : %s\n", st.enabled ? "Yes" : "No");
  fprintf(stderr, "AGC Type                  : %s\n", st.type == AGC_TYPE_LOWPASS ? "Lowpass" : "Harris");
  fprintf(stderr, "Blanking Counter          : %u ticks\n", st.blanking_counter);
  fprintf(stderr, "Blanking Limit            : %u ticks\n", st.blanking_limit);
  fprintf(stderr, "Lowpass Filter Coefficient: %0.3f\n", st.alpha);
  fprintf(stderr, "Deadband                  : %u dB\n", st.deadband_db);
  fprintf(stderr, "Operating Point           : %d dBFs\n", st.operating_point_dbfs);
  fprintf(stderr, "IF Gain                   : %u dB\n", st.rx_gain_db);
  fprintf(stderr, "/------------------------\n");
  fprintf(stderr, "Signal Magnitude          : %u\n", st.signal_magnitude);
  fprintf(stderr, "RSSI (Before Amp)         : %d dBFs\n", st.normalized_level_dbfs);
  fprintf(stderr, "/------------------------\n");
}

// ---- IQ dump tap ------------------------------------------------------------------------------------
void IqDataProcessor::enableIqDump(void) { iqDumpEnabled = true; }
void IqDataProcessor::disableIqDump(void) { iqDumpEnabled = false; }
bool IqDataProcessor::isIqDumpEnabled(void) { return iqDumpEnabled; }
void IqDataProcessor::registerIqDumpCallback(void (*cb)(int8_t *, uint32_t, void *), void *contextPtr)
{
  iqDumpContextPtr = contextPtr;
  iqDumpCallbackPtr = cb;
}

// ---- FrequencyScanner -------------------------------------------------------------------------------
FrequencyScanner::FrequencyScanner(IqDataProcessor *processorPtr) { this->processorPtr = processorPtr; }
FrequencyScanner::~FrequencyScanner(void) { stop(); }

bool FrequencyScanner::setScanParameters(uint64_t startFrequencyInHertz, uint64_t endFrequencyInHertz,
                                         uint64_t frequencyIncrementInHertz)
{
  return processorPtr->engine != 0 &&
         iqd_scanner_set_parameters(processorPtr->engine, 0, 1, startFrequencyInHertz, endFrequencyInHertz,
                                    frequencyIncrementInHertz) == IQD_OK;
}

bool FrequencyScanner::start(void)
{
  return processorPtr->engine != 0 && iqd_scanner_start(processorPtr->engine, 0, 1, 1) == IQD_OK;
}

bool FrequencyScanner::stop(void)
{
  return processorPtr->engine != 0 && iqd_scanner_start(processorPtr->engine, 0, 1, 0) == IQD_OK;
}

bool FrequencyScanner::isScanning(void)
{
  int scanning = 0;
  return processorPtr->engine != 0 && iqd_scanner_get(processorPtr->engine, 0, 0, 0, &scanning) == IQD_OK && scanning != 0;
}

uint64_t FrequencyScanner::getCurrentFrequencyInHertz(void)
{
  uint64_t hz = 0;
  if (processorPtr->engine != 0) iqd_scanner_get(processorPtr->engine, 0, &hz, 0, 0);
  return hz;
}

uint64_t FrequencyScanner::getTuneCount(void)
{
  uint64_t n = 0;
  if (processorPtr->engine != 0) iqd_scanner_get(processorPtr->engine, 0, 0, &n, 0);
  return n;
}
