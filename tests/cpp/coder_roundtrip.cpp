// Round-trip harness in the shape of the reference's only test, Test.cpp:15-118:
//   payload 'a'+i%26 -> encode -> AWGN (sd = 10^(-snr/20)) -> decode -> ErrNum / ThroughPut.
// Usage: coder_roundtrip <rate 0..5> <N> <srcBytes> <batch> <snr_dB> <SP|MS|CPU|TDMP|TDMPCL|MSCL|ENC> [seed]
//                        [--devices 0,0,...] [--host-input n] [--dump <prefix>] [--iters n]
// ENC: encoder only (no GPU): checks H c = 0 for every frame and prints "ParityFail=<n>".
// --devices: Coder::setDevices (one Coder over several HIP devices; an ordinal may repeat).
// --host-input: Coder::setHostInput (1 staged through the pinned ring = default, 2 page-lock the caller's pages).
// --dump: writes <prefix>.prior (encoded bytes), <prefix>.post (channel floats), <prefix>.out (decoded
//         bytes) so that a test can run the oracle on exactly these inputs.
// Prints NonZeros= and the reference's fields (sd=, Time=, <MODE>:<seconds>, ErrNum=, ThroughPut=).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <string>
#include <vector>

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
    srand(argc > 7 && argv[7][0] != '-' ? atoi(argv[7]) : 1);
    std::vector<int> devices;
    int hostInput = 0;
    std::string dump;
    int iters = 0;
    for (int i = 7; i + 1 < argc; ++i) {
        if (!strcmp(argv[i], "--host-input")) hostInput = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "--devices"))
            for (char *tok = strtok(argv[i + 1], ","); tok; tok = strtok(nullptr, ",")) devices.push_back(atoi(tok));
        else if (!strcmp(argv[i], "--dump")) dump = argv[i + 1];
        else if (!strcmp(argv[i], "--iters")) iters = atoi(argv[i + 1]);
    }
    auto save = [&](const char *ext, const void *p, size_t n) {
        if (dump.empty()) return;
        FILE *f = fopen((dump + ext).c_str(), "wb");
        if (f) { fwrite(p, 1, n, f); fclose(f); }
    };

    Coder coder(ldpcK, ldpcN, rate);
    cout << "NonZeros=" << coder.getNonZeros() << endl;
    if (iters > 0) coder.setMaxIterations(iters);
    if (!devices.empty()) coder.setDevices(devices.data(), (int)devices.size());
    if (hostInput) coder.setHostInput(hostInput);
    char *srcCode = (char *)malloc(srcLength);
    char *priorCode = (char *)malloc(coder.getPriorCodeLength(srcLength));
    float *postCode = (float *)malloc(sizeof(float) * coder.getPostCodeLength(srcLength));
    char *newSrcCode = (char *)calloc(srcLength + 1, 1);   // bytes no frame covers (K % 8 != 0) stay 0
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
            const long long at = (long long)f * ldpcK / 8;          // MyLdpc.cpp:559: offset * ldpcK / 8
            if (at < srcLength && memcmp(cw, srcCode + at, (size_t)std::min<long long>(ldpcK / 8, srcLength - at)))
                ++bad;   // systematic part must be the payload
        }
        cout << "ParityFail=" << bad << endl;
        save(".prior", priorCode, (size_t)coder.getPriorCodeLength(srcLength));
        if (!strcmp(mode, "ENC")) return bad ? 1 : 0;
    }
    if (coder.forDecoder(batch)) { cout << "forDecoder failed: " << coder.lastError() << endl; return 1; }
    const float sd = 1 / (pow(10, snr / 20));                          // Test.cpp:56
    cout << "sd=" << sd << endl;
    coder.test(priorCode, postCode, coder.getPriorCodeLength(srcLength), sd);
    save(".post", postCode, sizeof(float) * (size_t)coder.getPostCodeLength(srcLength));

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
    save(".out", newSrcCode, (size_t)srcLength);
    cout << "ErrNum=" << errNum << endl;
    cout << "ThroughPut=" << srcLength / decodeTime << endl;
    free(srcCode); free(priorCode); free(postCode); free(newSrcCode);
    return 0;
}
