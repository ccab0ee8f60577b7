// CoderBench -- what a user of the reference's class sees: the call sequence of Test.cpp:28-112
// (ctor, forEncoder, forDecoder, encode, test, addDecodeType, decode, compare) on any of the six
// 802.16e-seed codes at any size, every stage timed with the wall clock, decode() called several times
// on the same buffers.  bench.py runs it for the `coder_path` block of its JSON line:
//     CoderBench <rate 0..5> <N> <frames> <batch> <snr_dB> <SP|MS|CPU|TDMP|TDMPCL|MSCL>
//                [--iters n] [--repeat r] [--host-input 0|1|2] [--devices 0,1,...] [--seed s]
// Prints key=value lines; the reference's fields keep their names (sd=, Time=, ErrNum=, ThroughPut= in
// info bytes per second, Test.cpp:111; here for the best of the r calls, wall clock instead of clock()).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "MyLdpc.h"

static double now()
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}

int main(int argc, char **argv)
{
    if (argc < 7) {
        fprintf(stderr, "usage: CoderBench <rate 0..5> <N> <frames> <batch> <snr_dB> <mode> [--iters n] [--repeat r] "
                        "[--host-input m] [--devices a,b,...] [--seed s]\n");
        return 2;
    }
    const enum rate_type rate = (enum rate_type)atoi(argv[1]);
    const int ldpcN = atoi(argv[2]);
    static const int mbs[6] = {12, 8, 8, 6, 6, 4};
    if ((int)rate < 0 || (int)rate > 5 || ldpcN <= 0 || ldpcN % 24) return 2;
    const int ldpcK = ldpcN - mbs[(int)rate] * (ldpcN / 24);
    const long long frames = atoll(argv[3]);
    const int batch = atoi(argv[4]);
    const float snr = (float)atof(argv[5]);
    const char *mode = argv[6];
    int iters = 0, repeat = 3, hostInput = 0, seed = 1;
    std::vector<int> devices;
    for (int i = 7; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--iters")) iters = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--repeat")) repeat = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--host-input")) hostInput = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--seed")) seed = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--devices"))
            for (char *tok = strtok(argv[i + 1], ","); tok; tok = strtok(nullptr, ",")) devices.push_back(atoi(tok));
        else return 2;
    }
    const long long srcLL = frames * (ldpcK / 8);
    if (frames <= 0 || batch <= 0 || repeat <= 0 || srcLL > 0x7fffffffLL || frames * (long long)ldpcN > 0x7fffffffLL) {
        fprintf(stderr, "frames * K / 8 and frames * N must fit the reference's int lengths\n");
        return 2;
    }
    const int srcLength = (int)srcLL;
    srand((unsigned)seed);

    double t0 = now();
    Coder coder(ldpcK, ldpcN, rate);
    printf("K=%d\nN=%d\nz=%d\nNonZeros=%d\nframes=%lld\nbatch=%d\n", ldpcK, ldpcN, coder.getZ(), coder.getNonZeros(), frames, batch);
    if (iters > 0) coder.setMaxIterations(iters);
    if (!devices.empty()) coder.setDevices(devices.data(), (int)devices.size());
    if (hostInput) coder.setHostInput(hostInput);
    char *srcCode = (char *)malloc((size_t)srcLength);
    char *priorCode = (char *)malloc((size_t)coder.getPriorCodeLength(srcLength));
    float *postCode = (float *)malloc(sizeof(float) * (size_t)coder.getPostCodeLength(srcLength));
    char *newSrcCode = (char *)calloc((size_t)srcLength + 1, 1);
    if (!srcCode || !priorCode || !postCode || !newSrcCode) { fprintf(stderr, "out of host memory\n"); return 1; }
    for (int i = 0; i < srcLength; i++) srcCode[i] = (char)('a' + i % 26);          // Test.cpp:43-45

    t0 = now();
    if (coder.forEncoder()) { fprintf(stderr, "forEncoder: %s\n", coder.lastError()); return 1; }
    printf("forEncoder_s=%.6f\n", now() - t0);
    t0 = now();
    if (coder.encode(srcCode, priorCode, srcLength)) { fprintf(stderr, "encode: %s\n", coder.lastError()); return 1; }
    const double enc_s = now() - t0;
    printf("encode_s=%.6f\nencode_info_mbit_s=%.3f\n", enc_s, 8.0 * srcLength / enc_s / 1e6);
    t0 = now();
    if (coder.forDecoder(batch)) { fprintf(stderr, "forDecoder: %s\n", coder.lastError()); return 1; }
    printf("forDecoder_s=%.6f\n", now() - t0);
    const float sd = 1 / (pow(10, snr / 20));                                       // Test.cpp:56
    printf("sd=%g\n", sd);
    t0 = now();
    coder.test(priorCode, postCode, coder.getPriorCodeLength(srcLength), sd);
    printf("test_s=%.6f\n", now() - t0);

    enum decodeType t;
    if (!strcmp(mode, "SP")) t = DecodeSP;
    else if (!strcmp(mode, "MS")) t = DecodeMS;
    else if (!strcmp(mode, "CPU")) t = DecodeCPU;
    else if (!strcmp(mode, "TDMP")) t = DecodeTDMP;
    else if (!strcmp(mode, "TDMPCL")) t = DecodeTDMPCL;
    else if (!strcmp(mode, "MSCL")) t = DecodeMSCL;
    else return 2;
    t0 = now();
    if (coder.addDecodeType(t)) { fprintf(stderr, "addDecodeType: %s\n", coder.lastError()); return 1; }
    printf("addDecodeType_s=%.6f\n", now() - t0);
    double best = 1e30;
    for (int r = 0; r < repeat; ++r) {
        t0 = now();
        if (coder.decode(postCode, newSrcCode, srcLength, t)) { fprintf(stderr, "decode: %s\n", coder.lastError()); return 1; }
        const double dt = now() - t0;
        printf("decode_s[%d]=%.6f\n", r, dt);
        if (dt < best) best = dt;
    }
    printf("Time=%d\n", coder.lastIterations());
    printf("%s:%g\n", mode, best);
    long long errNum = 0;
    for (int i = 0; i < srcLength; ++i)
        if (srcCode[i] != newSrcCode[i]) ++errNum;                                  // Test.cpp:105-109
    printf("ErrNum=%lld\n", errNum);
    printf("ThroughPut=%g\n", srcLength / best);
    printf("decode_info_mbit_s=%.3f\n", 8.0 * srcLength / best / 1e6);
    free(srcCode); free(priorCode); free(postCode); free(newSrcCode);
    return 0;
}
