// iqdemod_file — file / pipe IQ source and S16_LE PCM sink around IqDataProcessor, the shape of
// the reference's radioApp.cc:103-111 (PCM to stdout) and demodulatorResearch/demodulators/demod.cc
// (samples from stdin).  Stands in for librtlsdr + DataConsumer when no dongle is attached:
//
//     iqdemod_file <mode 0-5> [threshold dBFS [agc type 0|1]] < capture_u8.iq | aplay -f S16_LE -r 8000
//
// With an AGC type the channel's AutomaticGainControl runs (Radio.cc:184: operating point -12 dBFS) and the
// IF gain it settles on is reported on stderr.
#include <stdio.h>
#include <stdlib.h>

#include "IqDataProcessor.h"

static void processPcmData(int16_t *bufferPtr, uint32_t bufferLength)
{
  fwrite(bufferPtr, sizeof(int16_t), bufferLength, stdout);
}

int main(int argc, char **argv)
{
  if (argc < 2) {
    fprintf(stderr, "usage: %s <mode: 0 none 1 am 2 fm 3 wbfm 4 lsb 5 usb> [squelch threshold dBFS [agc type: 0 lowpass 1 harris]]\n", argv[0]);
    return 2;
  }
  static char host[] = "127.0.0.1";
  IqDataProcessor processor(host, 8001);
  if (!processor.isOperational()) {
    fprintf(stderr, "iqdemod_file: %s\n", processor.lastError());
    return 1;
  }
  AmDemodulator am(processPcmData);
  FmDemodulator fm(processPcmData);
  WbFmDemodulator wbfm(processPcmData);
  SsbDemodulator ssb(processPcmData);
  processor.setAmDemodulator(&am);
  processor.setFmDemodulator(&fm);
  processor.setWbFmDemodulator(&wbfm);
  processor.setSsbDemodulator(&ssb);
  processor.setDemodulatorMode((IqDataProcessor::demodulatorType)atoi(argv[1]));
  if (argc > 2) processor.setSignalDetectThreshold(atoi(argv[2]));

  AutomaticGainControl agc(&processor, -12);
  if (argc > 3) {
    if (!agc.setType((uint32_t)atoi(argv[3]))) {
      fprintf(stderr, "iqdemod_file: invalid AGC type %s\n", argv[3]);
      return 2;
    }
    agc.enable();
  }

  static unsigned char block[32768];
  unsigned long timeStamp = 0;
  while (fread(block, 1, sizeof(block), stdin) == sizeof(block))
    processor.acceptIqData(timeStamp++, block, sizeof(block));
  fflush(stdout);
  if (argc > 3) fprintf(stderr, "IF gain: %u dB\n", agc.getReceiveIfGainInDb());
  return 0;
}
