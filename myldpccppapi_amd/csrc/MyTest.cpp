// MyTest -- command-line harness with the reference's interface (Test.cpp:15-118):
//     MyTest <srcBytes> <batch> <snr_dB> <SP|MS|CPU|TDMP|TDMPCL|MSCL>
// Same fixed code as the reference's test (z = 24, N = 576, rate 3/4B, Test.cpp:19-26), same
// payload ('a' + i % 26), same printed fields: sd=, Time=, <MODE>:<seconds>, ErrNum=, ThroughPut=
// (info bytes per second; wall-clock here, the reference prints CPU seconds from clock()).
// Differences: exits with status 2 on a usage error (the reference returns 0 silently,
// Test.cpp:16-17); the noise seed may be fixed with the environment variable MYTEST_SEED
// (the reference seeds from time(0), Test.cpp:29).
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
    if (argc != 5) {
        cerr << "usage: MyTest <srcBytes> <batch> <snr_dB> <SP|MS|CPU|TDMP|TDMPCL|MSCL>" << endl;
        return 2;
    }
    const int z = 24;
    const int ldpcN = z * 24;
    const int ldpcK = ldpcN / 4 * 3;
    const enum rate_type rate = rate_3_4_b;
    Coder coder(ldpcK, ldpcN, rate);
    const char *seed = getenv("MYTEST_SEED");
    srand(seed ? (unsigned)atoi(seed) : (unsigned)time(0));

    const int srcLength = atoi(argv[1]);
    if (srcLength <= 0 || atoi(argv[2]) <= 0) { cerr << "srcBytes and batch must be positive" << endl; return 2; }
    char *srcCode = (char *)malloc(srcLength);
    char *priorCode = (char *)malloc(coder.getPriorCodeLength(srcLength));
    float *postCode = (float *)malloc(sizeof(float) * (size_t)coder.getPostCodeLength(srcLength));
    char *newSrcCode = (char *)malloc(srcLength + 1);
    for (int i = 0; i < srcLength; i++) srcCode[i] = 'a' + i % 26;

    if (coder.forEncoder() || coder.forDecoder(atoi(argv[2])) || coder.encode(srcCode, priorCode, srcLength)) {
        cerr << coder.lastError() << endl;
        return 1;
    }
    const float snr = (float)atof(argv[3]);
    const float sd = 1 / (pow(10, snr / 20));
    cout << "sd=" << sd << endl;
    coder.test(priorCode, postCode, coder.getPriorCodeLength(srcLength), sd);

    static const struct { const char *name; enum decodeType t; } modes[] = {
        {"SP", DecodeSP}, {"MS", DecodeMS}, {"CPU", DecodeCPU}, {"TDMP", DecodeTDMP},
        {"TDMPCL", DecodeTDMPCL}, {"MSCL", DecodeMSCL}};
    int m = -1;
    for (int i = 0; i < 6; ++i)
        if (!strcmp(argv[4], modes[i].name)) m = i;
    if (m < 0) { cerr << "unknown mode " << argv[4] << endl; return 2; }
    if (coder.addDecodeType(modes[m].t)) { cerr << coder.lastError() << endl; return 1; }
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    if (coder.decode(postCode, newSrcCode, srcLength, modes[m].t)) { cerr << coder.lastError() << endl; return 1; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double decodeTime = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    cout << "Time=" << coder.lastIterations() << endl;
    cout << modes[m].name << ":" << decodeTime << endl;
    int errNum = 0;
    for (int i = 0; i < srcLength; ++i)
        if (srcCode[i] != newSrcCode[i]) ++errNum;
    cout << "ErrNum=" << errNum << endl;
    cout << "ThroughPut=" << srcLength / decodeTime << endl;
    free(srcCode); free(priorCode); free(postCode); free(newSrcCode);
    return 0;
}
