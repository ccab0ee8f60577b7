// Round-trip harness in the shape of the reference's only test, Test.cpp:15-118:
//   payload 'a'+i%26 -> encode -> AWGN (sd = 10^(-snr/20)) -> decode -> ErrNum / ThroughPut.
// Usage: coder_roundtrip <rate 0..5> <N> <srcBytes> <batch> <snr_dB> <SP|MS|CPU|TDMP|TDMPCL|MSCL|ENC> [seed]
// ENC: encoder only (no GPU): checks H c = 0 for every frame and prints "ParityFail=<n>".
// Prints the reference's fields (sd=, Time=, <MODE>:<seconds>, ErrNum=, ThroughPut=).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>

#include "MyLdpc.h"
using namespace std;

int main(int argc, char **argv)
{
    if (argc < 7) return 2;
    const enum rate_type rate = (enum rate_type)atoi(argv[1]);
    const int ldpcN = atoi(argv[2]);
    static const int mbs[6] = {12, 8, 8, 6, 6, 4};
    const int ldpcK = ldpcN - mbs[(int)rate] * (ldpcN / 24);
    const int srcLength = atoi(argv[3]);
    const int batch = atoi(argv[4]);
    const float snr = (float)atof(argv[5]);
    const char *mode = argv[6];
    srand(argc > 7 ? atoi(argv[7]) : 1);

    Coder coder(ldpcK, ldpcN, rate);
    char *srcCode = (char *)malloc(srcLength);
    char *priorCode = (char *)malloc(coder.getPriorCodeLength(srcLength));
    float *postCode = (float *)malloc(sizeof(float) * coder.getPostCodeLength(srcLength));
    char *newSrcCode = (char *)malloc(srcLength + 1);
    for (int i = 0; i < srcLength; i++) srcCode[i] = 'a' + i % 26;     // Test.cpp:43-45

    if (coder.forEncoder()) { cout << "forEncoder failed: " << coder.lastError() << endl; return 1; }
    if (coder.encode(srcCode, priorCode, srcLength)) { cout << "encode failed: " << coder.lastError() << endl; return 1; }
    {   // H c = 0 for every frame
        int bad = 0;
        const int frames = coder.getCodeSize(srcLength);
        const std::vector<int> &rr = coder.hRowRange(), &cc = coder.hCols();
        for (int f = 0; f < frames; ++f) {
            const unsigned char *cw = (const unsigned char *)priorCode + (size_t)f * (ldpcN / 8);
            for (size_t m = 0; m + 1 < rr.size(); ++m) {
                int par = 0;
                for (int p = rr[m]; p < rr[m + 1]; ++p) par ^= (cw[cc[p] / 8] >> (cc[p] % 8)) & 1;
                bad += par;
            }
            if (memcmp(cw, srcCode + (size_t)f * (ldpcK / 8),
                       (size_t)std::min(ldpcK / 8, srcLength - f * (ldpcK / 8))))
                ++bad;   // systematic part must be the payload
        }
        cout << "ParityFail=" << bad << endl;
        if (!strcmp(mode, "ENC")) return bad ? 1 : 0;
    }
    if (coder.forDecoder(batch)) { cout << "forDecoder failed: " << coder.lastError() << endl; return 1; }
    const float sd = 1 / (pow(10, snr / 20));                          // Test.cpp:56
    cout << "sd=" << sd << endl;
    coder.test(priorCode, postCode, coder.getPriorCodeLength(srcLength), sd);

    enum decodeType t;
    if (!strcmp(mode, "SP")) t = DecodeSP;
    else if (!strcmp(mode, "MS")) t = DecodeMS;
    else if (!strcmp(mode, "CPU")) t = DecodeCPU;
    else if (!strcmp(mode, "TDMP")) t = DecodeTDMP;
    else if (!strcmp(mode, "TDMPCL")) t = DecodeTDMPCL;
    else if (!strcmp(mode, "MSCL")) t = DecodeMSCL;
    else return 2;
    if (coder.addDecodeType(t)) { cout << "addDecodeType failed: " << coder.lastError() << endl; return 1; }
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    if (coder.decode(postCode, newSrcCode, srcLength, t)) { cout << "decode failed: " << coder.lastError() << endl; return 1; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double decodeTime = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    cout << "Time=" << coder.lastIterations() << endl;
    cout << mode << ":" << decodeTime << endl;
    int errNum = 0;
    for (int i = 0; i < srcLength; ++i)
        if (srcCode[i] != newSrcCode[i]) ++errNum;                     // Test.cpp:105-109
    cout << "ErrNum=" << errNum << endl;
    cout << "ThroughPut=" << srcLength / decodeTime << endl;
    free(srcCode); free(priorCode); free(postCode); free(newSrcCode);
    return 0;
}
