// C++ host surface of the engine with the reference's class and method names, so that code
// written against RtlSdrDiags' IqDataProcessor (hdr_diags/IqDataProcessor.h:16-58) and its four
// demodulator classes ({Am,Fm,WbFm,Ssb}Demodulator.h) compiles and behaves the same while
// the arithmetic runs on the GPU through the C ABI (include/iqdemod.h).
//
// One IqDataProcessor owns one single-channel engine.  Differences from the reference, all on
// the host side: acceptIqData() does not modify the caller's buffer (the converted bytes are
// available through the IQ dump tap); byteCount is whatever the read returned (Radio.cc:1895-1906
// forwards short reads), as long as it is a whole number of 64-byte units up to 32768 - anything
// else is counted, reported through lastError() and a line on stderr, never dropped silently; the
// IQ dump (IqDataProcessor.cc:756-760) hands its bytes to a callback instead of a UDP socket.
#pragma once
#include <stdint.h>

#include "iqdemod.h"

class IqDataProcessor;

// Common part of the four demodulator handles: a PCM sink and a gain, bound to an engine
// channel once the handle is given to an IqDataProcessor (set*Demodulator()).
class DemodulatorHandle
{
  public:
  typedef void (*PcmCallback)(int16_t *bufferPtr, uint32_t bufferLength);

  void resetDemodulator(void);
  void setDemodulatorGain(float gain);
  // {Am,Fm,WbFm,Ssb}Demodulator::acceptIqData (e.g. WbFmDemodulator.h:31, WbFmDemodulator.cc:383-411): signed bytes
  // straight into the demodulator, the PCM to its callback.  bufferLength: a multiple of 64 up to 32768; the buffer is
  // not modified.  A handle that no IqDataProcessor owns (the reference's offline harness,
  // demodulatorResearch/demodulators/demod.cc:262-285) runs on an engine of its own, made at the first call.
  void acceptIqData(int8_t *bufferPtr, uint32_t bufferLength);
  void displayInternalInformation(void);
  int lastStatusCode(void) const { return lastStatus; }
  ~DemodulatorHandle(void);

  protected:
  DemodulatorHandle(int demod, float defaultGain, PcmCallback pcmCallbackPtr);
  virtual void beforeAccept(void) {}

  int demod;                 // IQD_DEMOD_*
  float demodulatorGain;
  PcmCallback pcmCallbackPtr;
  iqd_t *engine;             // null until attached
  bool ownsEngine;           // made by acceptIqData() for a handle without a processor
  int lastStatus;
  int16_t pcmData[512];

  friend class IqDataProcessor;
};

class AmDemodulator : public DemodulatorHandle
{
  public:
  AmDemodulator(PcmCallback pcmCallbackPtr);      // AmDemodulator.h:24-26
};

class FmDemodulator : public DemodulatorHandle
{
  public:
  FmDemodulator(PcmCallback pcmCallbackPtr);      // FmDemodulator.h:24-26
};

class WbFmDemodulator : public DemodulatorHandle
{
  public:
  WbFmDemodulator(PcmCallback pcmCallbackPtr);    // WbFmDemodulator.h:24-26
};

class SsbDemodulator : public DemodulatorHandle
{
  public:
  SsbDemodulator(PcmCallback pcmCallbackPtr);     // SsbDemodulator.h:27-28
  void setLsbDemodulationMode(void);              // SsbDemodulator.h:30
  void setUsbDemodulationMode(void);              // SsbDemodulator.h:31

  private:
  virtual void beforeAccept(void);
  bool lsbDemodulationMode;
  friend class IqDataProcessor;
};

class IqDataProcessor
{
  public:
  enum demodulatorType {None = 0, Am = 1, Fm = 2, WbFm = 3, Lsb = 4, Usb = 5};

  IqDataProcessor(char *hostIpAddress, int hostPort);   // the UDP peer is accepted and ignored
  ~IqDataProcessor(void);

  void setDemodulatorMode(demodulatorType mode);
  void setAmDemodulator(AmDemodulator *demodulatorPtr);
  void setFmDemodulator(FmDemodulator *demodulatorPtr);
  void setWbFmDemodulator(WbFmDemodulator *demodulatorPtr);
  void setSsbDemodulator(SsbDemodulator *demodulatorPtr);
  void setSignalDetectThreshold(int32_t threshold);

  // hdr_diags/IqDataProcessor.h:32-33: in place, signed bytes, byteCount a multiple of 8
  void downconvertByFsOver4(int8_t *bufferPtr, uint32_t byteCount);
  void upconvertByFsOver4(int8_t *bufferPtr, uint32_t byteCount);

  void acceptIqData(unsigned long timeStamp, unsigned char *bufferPtr, unsigned long byteCount);

  void enableSignalNotification(void);
  void disableSignalNotification(void);
  void registerSignalStateCallback(void (*signalCallbackPtr)(bool signalPresent, void *contextPtr),
                                   void *contextPtr);
  void enableSignalMagnitudeNotification(void);
  void disableSignalMagnitudeNotification(void);
  void registerSignalMagnitudeCallback(void (*callbackPtr)(uint32_t signalMagnitude, void *contextPtr),
                                       void *contextPtr);

  // IQ dump tap (IqDataProcessor.h:54-56, IqDataProcessor.cc:756-760): the reference streams the rotated signed
  // bytes of every block to a UDP peer; here they go to a sink callback (no networking in this library).
  void enableIqDump(void);
  void disableIqDump(void);
  bool isIqDumpEnabled(void);
  void registerIqDumpCallback(void (*callbackPtr)(int8_t *bufferPtr, uint32_t byteCount, void *contextPtr),
                              void *contextPtr);

  // radio_adjustableReceiveGainInDb is a global in the reference (Radio.cc:19); here it is a
  // property of the processor.
  void setReceiveGainInDb(uint32_t gainInDb);

  void displayInternalInformation(void);
  bool isOperational(void) const { return engine != 0; }   // false when no HIP device was found
  const char *lastError(void) const;
  int lastStatusCode(void) const { return lastStatus; }     // IQD_OK or the failure of the most recent call
  unsigned long rejectedBlockCount(void) const { return rejectedBlocks; }   // acceptIqData calls that could not be processed
  // what the engine has queued on the device so far: kernel-launch calls and copy / fill operations (diagnostics)
  void deviceOperationCounts(unsigned long long *launches, unsigned long long *copies) const;

  private:
  friend class AutomaticGainControl;
  friend class FrequencyScanner;
  void attach(DemodulatorHandle *h);

  iqd_t *engine;
  demodulatorType demodulatorMode;
  int32_t signalDetectThreshold;
  uint32_t blockBytes;
  AmDemodulator *amDemodulatorPtr;
  FmDemodulator *fmDemodulatorPtr;
  WbFmDemodulator *wbFmDemodulatorPtr;
  SsbDemodulator *ssbDemodulatorPtr;
  bool signalNotificationEnabled;
  void *signalCallbackContextPtr;
  void (*signalCallbackPtr)(bool signalPresent, void *contextPtr);
  bool signalMagnitudeNotificationEnabled;
  void *signalMagnitudeCallbackContextPtr;
  void (*signalMagnitudeCallbackPtr)(uint32_t signalMagnitude, void *contextPtr);
  bool iqDumpEnabled;
  void *iqDumpContextPtr;
  void (*iqDumpCallbackPtr)(int8_t *bufferPtr, uint32_t byteCount, void *contextPtr);
  int16_t pcmData[512];      // one block's PCM (32768 / 64)
  int8_t dumpData[32768];
  int lastStatus;
  unsigned long receiveBlockCount;
  unsigned long rejectedBlocks;
};

// hdr_diags/AutomaticGainControl.h:22-47.  The reference's constructor takes its owning Radio and reaches the
// processor through it (AutomaticGainControl.cc:170-186); here the processor is handed over directly, and the
// IF gain the AGC moves is the processor's (Radio::get/setReceiveIfGainInDb).  The AGC itself runs on the GPU,
// once per accepted block, inside iqd_accept_iq.
#define AGC_TYPE_LOWPASS (0)
#define AGC_TYPE_HARRIS (1)

class AutomaticGainControl
{
  public:
  AutomaticGainControl(IqDataProcessor *processorPtr, int32_t operatingPointInDbFs);
  ~AutomaticGainControl(void);

  void setOperatingPoint(int32_t operatingPointInDbFs);
  bool setAgcFilterCoefficient(float coefficient);
  bool setType(uint32_t type);
  bool setDeadband(uint32_t deadbandInDb);
  bool setBlankingLimit(uint32_t blankingLimit);
  bool enable(void);
  bool disable(void);
  bool isEnabled(void);
  uint32_t getSignalMagnitude(void);
  uint32_t getReceiveIfGainInDb(void);        // Radio::getReceiveIfGainInDb, Radio.cc:1223-1229
  void displayInternalInformation(void);

  private:
  IqDataProcessor *processorPtr;
};

// hdr_diags/FrequencyScanner.h:18-75.  The reference's constructor takes the Radio; here the processor.  The
// scanner runs on the GPU inside iqd_accept_iq (one step per block the squelch rejects); the frequency it wants
// the tuner on is read back with getCurrentFrequencyInHertz() - what Radio::setReceiveFrequency was told last.
class FrequencyScanner
{
  public:
  FrequencyScanner(IqDataProcessor *processorPtr);
  ~FrequencyScanner(void);

  bool setScanParameters(uint64_t startFrequencyInHertz, uint64_t endFrequencyInHertz,
                         uint64_t frequencyIncrementInHertz);
  bool start(void);
  bool stop(void);
  bool isScanning(void);
  uint64_t getCurrentFrequencyInHertz(void);
  uint64_t getTuneCount(void);

  private:
  IqDataProcessor *processorPtr;
};
