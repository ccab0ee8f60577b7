/*
 * MyLdpc.h -- the reference's public C++ API (class Coder, MyLdpc.h:104-238 of
 * wing02/MyLdpcCppApi) re-created on top of the MI355X C ABI (ldpc_hip.h).
 *
 * Same names, argument meaning, call order and data conventions as the
 * reference (Test.cpp:28-104 compiles against this header unchanged apart from
 * its `#include "cl.hpp"`):
 *     Coder c(K, N, rate); c.forEncoder(); c.forDecoder(batch);
 *     c.encode(src, prior, srcLen); c.test(prior, post, priorLen, sd);
 *     c.addDecodeType(DecodeSP); c.decode(post, out, srcLen, DecodeSP);
 *
 * Deliberate differences (all listed in INTEGRATION.md):
 *   - no Eigen, no OpenCL: the public Eigen member `checkMatrix` (MyLdpc.h:128)
 *     becomes the CSR view hRowRange()/hCols(); `kernelSourceCode` and the
 *     run-time dependency on ./decodeCL.c (MyLdpc.cpp:237) are gone.
 *   - methods return 0 on success as before, but failures return a non-zero
 *     ldpc_status and never exit() (reference: MyLdpc.cpp:243-254); lastError().
 *   - several decode types may be added to one Coder (the reference shares its
 *     kernel objects between SP/MS/TDMP, MyLdpc.cpp:334,397,450).
 *   - DecodeCPU runs the same min-sum arithmetic on the GPU (bit-identical to
 *     decodeCPU, MyLdpc.cpp:684-784, including its bit-offset packing); there is
 *     no CPU decode path in this library.
 *   - DecodeTDMPCL runs the layered schedule with the semantics of the fused kernel
 *     (decodeCL.c:307-426).  DecodeTDMP follows the reference's HOST-driven layered path
 *     (MyLdpc.cpp:889-976 over decodeCL.c:203-300) operation for operation on the seeds where
 *     that path is a decode of H -- every row of one weight: rates 2/3A and 5/6; the reference
 *     sizes layer l as hRowRange[l+z]-hRowRange[l], :907,958 -- and falls back to the fused
 *     kernel's semantics on the other four.  DecodeMSCL runs the fused flooding kernel's
 *     arithmetic (decodeCL.c:432-567, 120 iterations as there).
 *   - `times` (MyLdpc.cpp:24) and the SP channel scale 8 (decodeCL.c:9) stay the
 *     defaults and can be changed with setMaxIterations()/setLlrScale().
 *   - setDevices(): one Coder over several GPUs (the reference uses devices[0] only).
 */
#ifndef MYLDPC_H_
#define MYLDPC_H_

#include <map>
#include <string>
#include <vector>

#include "ldpc_hip.h"

#define LDPC_SUCCESS 0
#define LDPC_FAIL 1

enum rate_type { rate_1_2, rate_2_3_a, rate_2_3_b, rate_3_4_a, rate_3_4_b, rate_5_6 };

enum decodeType { DecodeCPU, DecodeMS, DecodeSP, DecodeTDMP, DecodeTDMPCL, DecodeMSCL };

const int n_b = 24;

class Coder {
public:
    Coder(int ldpcK, int ldpcN, enum rate_type rate);
    ~Coder();
    Coder(const Coder &) = delete;
    Coder &operator=(const Coder &) = delete;

    int forEncoder();
    int forDecoder(int batchSize);
    int addDecodeType(enum decodeType deType);

    int encode(char *srcCode, char *priorCode, int srcLength);
    /* postCode: getPostCodeLength(srcLength) floats; srcCode: srcLength bytes out */
    int decode(float *postCode, char *srcCode, int srcLength, enum decodeType deType);

    /* BPSK + AWGN of standard deviation `rate` on libc rand() (MyLdpc.cpp:1061-1105) */
    int test(char *priorCode, float *postCode, int priorCodeLength, float rate);

    int getPriorCodeLength(int srcLength);
    int getPostCodeLength(int srcLength);
    int getCodeSize(int srcLength);

    /* ---- additions ------------------------------------------------------ */
    void setMaxIterations(int times) { this->times = times; }   /* before addDecodeType */
    void setLlrScale(float s) { llrScale = s; }
    void setDevice(int ordinal) { device = ordinal; devices.clear(); }
    /* Several HIP devices behind one Coder (the reference opens devices[0] only, MyLdpc.cpp:235):
     * decode() then cuts the frame stream into one contiguous range per entry and decodes the
     * ranges side by side, batchSize frames per device and launch group; bytes identical to the
     * single-device result.  Before addDecodeType(). */
    void setDevices(const int *ordinals, int count) { devices.assign(ordinals, ordinals + (count > 0 ? count : 0)); }
    /* How decode() moves the caller's (pageable) postCode to the devices: LDPC_HOST_INPUT_STAGED (default:
     * through the library's own pinned ring) or LDPC_HOST_INPUT_LOCK_PAGES (page-locks the caller's pages
     * for the call; include/ldpc_hip.h).  Before addDecodeType(). */
    void setHostInput(int mode) { hostInput = mode; }
    int lastIterations() const { return lastTime; }             /* the reference's "Time=" */
    const char *lastError() const { return err.c_str(); }
    int getNonZeros() const { return nonZeros; }
    int getZ() const { return z; }
    /* H in CSR form, row-major edge order (replaces the Eigen checkMatrix member) */
    const std::vector<int> &hRowRange() const { return rowRange; }
    const std::vector<int> &hRows() const { return rows; }
    const std::vector<int> &hCols() const { return cols; }

private:
    int initCheckMatrix();
    int encodeOnce(const char *src, char *code, int srcLength);
    int fail(int code, const std::string &msg);

    int times;
    float llrScale;
    int device;
    std::vector<int> devices;        /* setDevices(); empty: `device` alone */
    int hostInput = 0;               /* setHostInput() */
    int makeDecoder(const ldpc_decoder_config &cfg, ldpc_decoder **out);
    const signed char *hSeed;
    int seedRowLength;
    int ldpcK, ldpcN, ldpcM, z, nonZeros, batchSize;
    enum rate_type rate;
    bool isEncoder, isDecoder;
    int lastTime;
    std::string err;

    std::vector<int> rows, cols, rowRange;

    /* encoder (structured, replaces the dense Richardson-Urbanke precompute) */
    std::vector<int> shift;          /* [seedRowLength*24] scaled shifts, -1 = empty */
    int encX;                        /* block row of the weight-3 parity column's middle entry */
    std::vector<unsigned char> denseInv; /* fallback: bit-packed inverse of the parity part */
    bool structured;

    ldpc_graph *graph;
    std::map<int, ldpc_decoder *> decoders; /* decodeType -> handle */
    int cpuDecoderBatch;
};

float gaussian(float ave, float sd);

#endif /* MYLDPC_H_ */
