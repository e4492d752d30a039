// iqdemod_file — file / pipe IQ source and S16_LE PCM sink around IqDataProcessor, the shape of
// the reference's radioApp.cc:103-111 (PCM to stdout) and demodulatorResearch/demodulators/demod.cc
// (samples from stdin).  Stands in for librtlsdr + DataConsumer when no dongle is attached:
//
//     iqdemod_file <mode 0-5> [threshold dBFS [agc type 0|1]] < capture_u8.iq | aplay -f S16_LE -r 8000
//
// With an AGC type the channel's AutomaticGainControl runs (Radio.cc:184: operating point -12 dBFS) and the
// IF gain it settles on is reported on stderr.  Further options, anywhere after the mode:
//     scan=<start>:<end>:<increment>   run the channel's FrequencyScanner (Hz); its last tuning command is reported
//     dump=<file>                      IQ dump tap: the rotated signed bytes of every block go to <file>
//     blocks=<n1,n2,...>               read blocks of these sizes in turn (default 32768): short reads
//     timing=<file>                    wall time of every acceptIqData call (microseconds, one per line) and the
//                                      device operations the engine queued, for bench.py --config 0
//     demod                            the reference's offline harness instead (demodulatorResearch/demodulators/
//                                      demod.cc:210-290): stdin holds SIGNED bytes, which go straight into a bare
//                                      demodulator object - {Am,Fm,WbFm,Ssb}Demodulator::acceptIqData, no processor,
//                                      no squelch, 16384 bytes per read like the reference.  Types 4 AND 5 demodulate the
//                                      UPPER sideband, as the reference program does: its switch has no break (demod.cc:232-242),
//                                      so setLsbDemodulationMode() is followed by setUsbDemodulationMode().  Pinned by
//                                      tests/golden/demod_tool.npz, the reference program's own output.
//     sideband=lsb                     with `demod`: really the lower sideband (what -d 4 was meant to select)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <vector>

#include "IqDataProcessor.h"

static void processPcmData(int16_t *bufferPtr, uint32_t bufferLength)
{
  fwrite(bufferPtr, sizeof(int16_t), bufferLength, stdout);
}

static void dumpIqData(int8_t *bufferPtr, uint32_t byteCount, void *contextPtr)
{
  fwrite(bufferPtr, 1, byteCount, (FILE *)contextPtr);
}

int main(int argc, char **argv)
{
  if (argc < 2) {
    fprintf(stderr, "usage: %s <mode: 0 none 1 am 2 fm 3 wbfm 4 lsb 5 usb> [squelch threshold dBFS [agc type: 0 lowpass 1 harris]]\n", argv[0]);
    return 2;
  }
  const char *scanSpec = 0, *dumpPath = 0, *blockSpec = 0, *timingPath = 0;
  bool demodOnly = false, forceLsb = false;
  int npos = 0;
  char *pos[8];
  for (int i = 1; i < argc && npos < 8; i++) {
    if (strncmp(argv[i], "scan=", 5) == 0) scanSpec = argv[i] + 5;
    else if (strncmp(argv[i], "dump=", 5) == 0) dumpPath = argv[i] + 5;
    else if (strncmp(argv[i], "blocks=", 7) == 0) blockSpec = argv[i] + 7;
    else if (strncmp(argv[i], "timing=", 7) == 0) timingPath = argv[i] + 7;
    else if (strcmp(argv[i], "demod") == 0) demodOnly = true;
    else if (strcmp(argv[i], "sideband=lsb") == 0) forceLsb = true;
    else pos[npos++] = argv[i];
  }
  argc = npos + 1;
  for (int i = 0; i < npos; i++) argv[i + 1] = pos[i];
  static unsigned char block[32768];
  size_t sizes[64], nsizes = 0, next = 0;
  if (blockSpec != 0)
    for (const char *q = blockSpec; *q != 0 && nsizes < 64;) {
      sizes[nsizes++] = (size_t)strtoul(q, (char **)&q, 10);
      if (*q == ',') q++;
    }
  if (demodOnly) {   // demod.cc: one demodulator object by itself, signed samples in, PCM out
    AmDemodulator amAlone(processPcmData);
    FmDemodulator fmAlone(processPcmData);
    WbFmDemodulator wbfmAlone(processPcmData);
    SsbDemodulator ssbAlone(processPcmData);
    const int type = atoi(argv[1]);
    if (type < 1 || type > 5) { fprintf(stderr, "iqdemod_file: demodulator type 1-5\n"); return 2; }
    switch (type) {   // (the reference's switch, fall-through included: demod.cc:232-242)
    case 4: ssbAlone.setLsbDemodulationMode();   // fall through
    case 5: ssbAlone.setUsbDemodulationMode();
    }
    if (forceLsb) ssbAlone.setLsbDemodulationMode();
    DemodulatorHandle *d = type == 1 ? (DemodulatorHandle *)&amAlone : type == 2 ? (DemodulatorHandle *)&fmAlone
                         : type == 3 ? (DemodulatorHandle *)&wbfmAlone : (DemodulatorHandle *)&ssbAlone;
    for (;;) {
      size_t want = nsizes ? sizes[next++ % nsizes] : 16384;   // (demod.cc:249: fread(inputBuffer, sizeof(int8_t), 16384, stdin))
      if (want == 0 || want > sizeof(block)) want = sizeof(block);
      const size_t got = fread(block, 1, want, stdin);
      if (got == 0) break;
      d->acceptIqData((int8_t *)block, (uint32_t)got);
      if (d->lastStatusCode() != IQD_OK) return 3;
      if (got < want) break;
    }
    fflush(stdout);
    return 0;
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

  FrequencyScanner scanner(&processor);
  if (scanSpec != 0) {
    unsigned long long a = 0, b = 0, inc = 0;
    if (sscanf(scanSpec, "%llu:%llu:%llu", &a, &b, &inc) != 3 || !scanner.setScanParameters(a, b, inc) || !scanner.start()) {
      fprintf(stderr, "iqdemod_file: bad scan specification %s\n", scanSpec);
      return 2;
    }
  }
  FILE *dumpFile = 0;
  if (dumpPath != 0) {
    dumpFile = fopen(dumpPath, "wb");
    if (dumpFile == 0) { perror(dumpPath); return 2; }
    processor.registerIqDumpCallback(dumpIqData, dumpFile);
    processor.enableIqDump();
  }

  // Blocks of 32768 bytes like Radio.cc:1895; a read that comes back short (a pipe, the end of the file) is handed
  // on as it is, like Radio.cc:1895-1906 does - the processor takes whole 64-byte units and reports the rest.
  // Extra option blocks=<a,b,c,...>: read these block sizes in turn instead (short-read experiments).
  unsigned long timeStamp = 0;
  std::vector<double> blockMicroseconds;
  for (;;) {
    size_t want = nsizes ? sizes[next++ % nsizes] : sizeof(block);
    if (want == 0 || want > sizeof(block)) want = sizeof(block);
    const size_t got = fread(block, 1, want, stdin);
    if (got == 0) break;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    processor.acceptIqData(timeStamp++, block, got);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (timingPath != 0) blockMicroseconds.push_back(1e6 * (double)(t1.tv_sec - t0.tv_sec) + 1e-3 * (double)(t1.tv_nsec - t0.tv_nsec));
    if (got < want) break;
  }
  fflush(stdout);
  if (timingPath != 0) {
    FILE *tf = fopen(timingPath, "w");
    if (tf == 0) { perror(timingPath); return 2; }
    unsigned long long launches = 0, copies = 0;
    processor.deviceOperationCounts(&launches, &copies);
    fprintf(tf, "blocks %zu launches %llu copies %llu\n", blockMicroseconds.size(), launches, copies);
    for (size_t i = 0; i < blockMicroseconds.size(); i++) fprintf(tf, "%.3f\n", blockMicroseconds[i]);
    fclose(tf);
  }
  if (processor.rejectedBlockCount() != 0)
    fprintf(stderr, "iqdemod_file: %lu block(s) could not be processed (%s)\n", processor.rejectedBlockCount(), processor.lastError());
  if (argc > 3) fprintf(stderr, "IF gain: %u dB\n", agc.getReceiveIfGainInDb());
  if (scanSpec != 0)
    fprintf(stderr, "scanner: %llu Hz after %llu tuning commands\n",
            (unsigned long long)scanner.getCurrentFrequencyInHertz(), (unsigned long long)scanner.getTuneCount());
  if (dumpFile != 0) fclose(dumpFile);
  return processor.rejectedBlockCount() != 0 ? 3 : 0;
}
