/*
 * MyLdpc.cpp -- class Coder of wing02/MyLdpcCppApi re-created above the MI355X
 * C ABI (include/ldpc_hip.h).  Host-side C++ only: every decode runs in the HIP
 * kernels of libldpc_hip.so.  Reference lines are cited per method.
 */
#include "../../include/MyLdpc.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "wimax_seeds.h"

namespace {
const int kSeedCols = WIMAX_NB;
}

int Coder::fail(int code, const std::string &msg)
{
    err = msg;
    return code ? code : LDPC_FAIL;
}

/* MyLdpc.cpp:20-29 */
Coder::Coder(int ldpcK, int ldpcN, enum rate_type rate)
    : times(40), llrScale(8.0f), device(0), hSeed(nullptr), seedRowLength(0), ldpcK(ldpcK),
      ldpcN(ldpcN), ldpcM(ldpcN - ldpcK), z(0), nonZeros(0), batchSize(0), rate(rate),
      isEncoder(false), isDecoder(false), lastTime(0), encX(-1), structured(false), graph(nullptr),
      cpuDecoderBatch(0)
{
    initCheckMatrix();
}

Coder::~Coder()
{
    for (auto &kv : decoders) ldpc_decoder_destroy(kv.second);
    if (graph) ldpc_graph_destroy(graph);
}

/* MyLdpc.cpp:52-135.  The triplet list + Eigen setFromTriplets become a direct
 * row-major listing: within a row the blocks come in ascending block column. */
int Coder::initCheckMatrix()
{
    z = ldpcN / n_b;                                        /* :55 */
    const int r = (int)rate;
    if (r < 0 || r >= WIMAX_NUM_RATES || z <= 0 || z * n_b != ldpcN)
        return fail(LDPC_ERR_ARG, "Coder: N must be a positive multiple of 24 and rate one of the six seeds");
    hSeed = wimax_seed_table[r];
    seedRowLength = wimax_seed_rows[r];                     /* :58-83 */
    if (ldpcM != seedRowLength * z)
        return fail(LDPC_ERR_ARG, "Coder: N-K does not match the rate's seed matrix");
    shift.assign((size_t)seedRowLength * kSeedCols, -1);
    for (int i = 0; i < seedRowLength * kSeedCols; ++i) {
        int p = hSeed[i];
        if (p >= 0) shift[i] = (rate != rate_2_3_a) ? p * z / WIMAX_Z0 : p % z;   /* :89-94 */
    }
    rows.clear(); cols.clear();
    rowRange.assign((size_t)ldpcM + 1, 0);
    for (int sr = 0; sr < seedRowLength; ++sr)
        for (int pr = 0; pr < z; ++pr) {
            rowRange[(size_t)sr * z + pr] = (int)rows.size();
            for (int sc = 0; sc < kSeedCols; ++sc) {
                const int p = shift[(size_t)sr * kSeedCols + sc];
                if (p < 0) continue;
                rows.push_back(sr * z + pr);                /* (z + c - r) % z == p, :97 */
                cols.push_back(sc * z + (pr + p) % z);
            }
        }
    rowRange[ldpcM] = (int)rows.size();
    nonZeros = (int)rows.size();                            /* :109 */
    return LDPC_SUCCESS;
}

/* Length helpers, MyLdpc.cpp:620-631. */
int Coder::getPriorCodeLength(int srcLength) { return (srcLength + (ldpcK / 8) - 1) / (ldpcK / 8) * (ldpcN / 8); }
int Coder::getPostCodeLength(int srcLength) { return (srcLength + (ldpcK / 8) - 1) / (ldpcK / 8) * ldpcN; }
int Coder::getCodeSize(int srcLength) { return (srcLength + (ldpcK / 8) - 1) / (ldpcK / 8); }

/* ----------------------------------------------------------------- encoder */

/* forEncoder, MyLdpc.cpp:137-165.  The reference inverts dense (M-z)^2 int
 * matrices (Richardson-Urbanke with gap z), which cannot scale past a few
 * thousand bits.  A systematic codeword is unique once H's parity part is
 * nonsingular, so any exact solver of H [s p]^T = 0 gives the same bytes.  The
 * 802.16e seeds have a weight-3 parity block column (rows 0, x, mb-1; equal
 * outer shifts, middle shift 0) followed by a dual diagonal, which solves in
 * O(E): p1 = sum of all block rows of A s, then forward substitution. */
int Coder::forEncoder()
{
    if (!hSeed) return fail(LDPC_ERR_STATE, "Coder was not constructed");
    const int mb = seedRowLength, kb = kSeedCols - mb;
    structured = true;
    encX = -1;
    int cnt = 0;
    for (int i = 0; i < mb; ++i) {
        const int p = shift[(size_t)i * kSeedCols + kb];
        if (p < 0) continue;
        ++cnt;
        if (i != 0 && i != mb - 1) { if (p != 0 || encX >= 0) structured = false; encX = i; }
    }
    if (cnt != 3 || encX < 0 || shift[kb] < 0 || shift[(size_t)(mb - 1) * kSeedCols + kb] != shift[kb])
        structured = false;
    for (int j = 0; j < mb - 1 && structured; ++j)
        for (int i = 0; i < mb; ++i) {
            const int p = shift[(size_t)i * kSeedCols + kb + 1 + j];
            const bool want = (i == j || i == j + 1);
            if (want ? (p != 0) : (p >= 0)) { structured = false; break; }
        }
    if (!structured) {
        /* generic fallback: bit-packed Gauss-Jordan inverse of the M x M parity part */
        const int M = ldpcM;
        if (M > 8192) return fail(LDPC_ERR_UNSUPPORTED, "forEncoder: unstructured parity part too large");
        const size_t W = ((size_t)2 * M + 63) / 64;
        std::vector<unsigned long long> a((size_t)M * W, 0ull);
        for (int e = 0; e < nonZeros; ++e)
            if (cols[e] >= ldpcK) {
                const int c = cols[e] - ldpcK;
                a[(size_t)rows[e] * W + c / 64] |= 1ull << (c % 64);
            }
        for (int i = 0; i < M; ++i) a[(size_t)i * W + (M + i) / 64] |= 1ull << ((M + i) % 64);
        for (int c = 0; c < M; ++c) {
            int piv = -1;
            for (int r2 = c; r2 < M; ++r2)
                if (a[(size_t)r2 * W + c / 64] >> (c % 64) & 1ull) { piv = r2; break; }
            if (piv < 0) return fail(LDPC_ERR_UNSUPPORTED, "forEncoder: parity part of H is singular");
            if (piv != c)
                for (size_t w = 0; w < W; ++w) std::swap(a[(size_t)piv * W + w], a[(size_t)c * W + w]);
            for (int r2 = 0; r2 < M; ++r2)
                if (r2 != c && (a[(size_t)r2 * W + c / 64] >> (c % 64) & 1ull))
                    for (size_t w = 0; w < W; ++w) a[(size_t)r2 * W + w] ^= a[(size_t)c * W + w];
        }
        denseInv.assign((size_t)M * M, 0);
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < M; ++j)
                denseInv[(size_t)i * M + j] = (unsigned char)(a[(size_t)i * W + (M + j) / 64] >> ((M + j) % 64) & 1ull);
    }
    isEncoder = true;
    return LDPC_SUCCESS;
}

/* encode, MyLdpc.cpp:554-569, index for index: frame f reads srcCode + f*K/8 (integer division
 * of the PRODUCT, as the decoders write frame f at (f*K)/8, decodeCL.c:191-192) and writes
 * priorCode + f*N/8; the last frame may be short (zero-padded, :639-650).  For K % 8 != 0 --
 * e.g. Coder(324, 648, rate_1_2) -- frames therefore start at bytes 0, 40, 81, 121, ...: every
 * other frame skips one source byte, exactly as in the reference. */
int Coder::encode(char *srcCode, char *priorCode, int srcLength)
{
    if (!isEncoder) return fail(LDPC_ERR_STATE, "encode: call forEncoder() first");
    if (!srcCode || !priorCode || srcLength <= 0) return fail(LDPC_ERR_ARG, "encode: bad arguments");
    /* frames [0, last]: frame `offset` starts at byte offset*K/8 and is the last one once (offset+1)*K/8 >= srcLength
     * (:557-565).  The frames are independent and encodeOnce only reads the object: large payloads are spread over host
     * threads, contiguous frame ranges each (4096 frames of the (64800, 32400) code: 41 ms on one core). */
    long long last = 0;
    while ((last + 1) * (long long)ldpcK / 8 < srcLength) ++last;
    auto one = [&](long long offset) {
        const int at = (int)(offset * ldpcK / 8);
        encodeOnce(&srcCode[at], &priorCode[offset * ldpcN / 8], offset < last ? ldpcK / 8 : srcLength - at);
    };
    const long long frames = last + 1;
    const unsigned hw = std::thread::hardware_concurrency();
    const long long threads = std::min<long long>(std::min<long long>(hw ? hw : 1, 16), frames * (long long)ldpcN / (1 << 21));
    if (threads <= 1) {
        for (long long offset = 0; offset < frames; ++offset) one(offset);
        return LDPC_SUCCESS;
    }
    std::vector<std::thread> pool;
    for (long long t = 0; t < threads; ++t) {
        const long long lo = frames * t / threads, hi = frames * (t + 1) / threads;
        pool.emplace_back([&one, lo, hi] { for (long long offset = lo; offset < hi; ++offset) one(offset); });
    }
    for (auto &th : pool) th.join();
    return LDPC_SUCCESS;
}

/* ---- z-bit vectors as 64-bit words (bit r of a block = bit r % 64 of word r / 64), each followed by one
 *      zero word so that a 64-bit window may start at any bit ---------------------------------------- */
namespace {

inline unsigned long long window64(const unsigned long long *v, long long bit)
{
    const long long w = bit >> 6;
    const int b = (int)(bit & 63);
    return b ? (v[w] >> b) | (v[w + 1] << (64 - b)) : v[w];
}

/* acc[dst .. dst + count) ^= v[src .. src + count) */
inline void xor_bits(unsigned long long *acc, long long dst, const unsigned long long *v, long long src, long long count)
{
    while (count > 0) {
        const long long dw = dst >> 6;
        const int db = (int)(dst & 63);
        const int n = (int)(count < 64 - db ? count : 64 - db);
        const unsigned long long bits = window64(v, src) & (n == 64 ? ~0ull : ((1ull << n) - 1ull));
        acc[dw] ^= bits << db;
        dst += n; src += n; count -= n;
    }
}

/* acc[r] ^= v[(r + p) mod z] for r < z: the action of a circulant with shift p (row r has its one at
 * column (r + p) mod z, initCheckMatrix above) */
inline void xor_rotated(unsigned long long *acc, const unsigned long long *v, int p, int z)
{
    xor_bits(acc, 0, v, p, z - p);
    if (p) xor_bits(acc, z - p, v, 0, p);
}

}  // namespace

/* encodeOnce, MyLdpc.cpp:633-682.  Output layout [K info | z p1 | M-z p2],
 * LSB-first bits (:661-680).  The structured solve works on whole z-bit vectors, 64 bits per operation
 * (the bit-at-a-time form it replaces spent 0.29 ms per frame at N = 64800; this one about 10 us); the
 * dense fallback for an unstructured parity part stays bit-wise. */
int Coder::encodeOnce(const char *src, char *code, int srcLength)
{
    const int mb = seedRowLength, kb = kSeedCols - mb;
    if (srcLength < 0) srcLength = 0;
    if (srcLength > ldpcK / 8) srcLength = ldpcK / 8;
    /* the reference copies the source with strncpy (:661), which stops at a NUL
     * byte; the intent -- and this code -- is a plain copy */
    memcpy(code, src, (size_t)srcLength);
    memset(code + srcLength, 0, (size_t)(ldpcN / 8 - srcLength));
    if (structured) {
        const int W = (z + 63) / 64 + 1;                      /* words per block incl. the zero word */
        /* information bits: the first K/8 whole bytes of the frame (:641-650), zero beyond srcLength -- exactly
         * the bytes just written to `code`; as words with slack for the 64-bit windows */
        const size_t infoWords = ((size_t)ldpcK + 63) / 64 + 2;
        std::vector<unsigned long long> info(infoWords, 0ull), lam((size_t)mb * W, 0ull), par((size_t)mb * W, 0ull), blk((size_t)W, 0ull);
        memcpy(info.data(), code, (size_t)srcLength);
        /* lambda_i = (A s) restricted to block row i */
        for (int j = 0; j < kb; ++j) {
            std::fill(blk.begin(), blk.end(), 0ull);
            xor_bits(blk.data(), 0, info.data(), (long long)j * z, z);
            bool any = false;
            for (int w = 0; w + 1 < W; ++w) any = any || blk[w];
            if (!any) continue;
            for (int i = 0; i < mb; ++i) {
                const int p = shift[(size_t)i * kSeedCols + j];
                if (p >= 0) xor_rotated(&lam[(size_t)i * W], blk.data(), p, z);
            }
        }
        unsigned long long *p1 = &par[0];
        for (int i = 0; i < mb; ++i)
            for (int w = 0; w + 1 < W; ++w) p1[w] ^= lam[(size_t)i * W + w];
        const int h0 = shift[kb];
        /* v_1 = lambda_0 + P(h0) p1 ; v_{i+1} = lambda_i + v_i (+ p1 at row x) */
        for (int w = 0; w + 1 < W; ++w) par[(size_t)W + w] = lam[w];
        xor_rotated(&par[(size_t)W], p1, h0, z);
        for (int i = 1; i <= mb - 2; ++i)
            for (int w = 0; w + 1 < W; ++w)
                par[(size_t)(i + 1) * W + w] = lam[(size_t)i * W + w] ^ par[(size_t)i * W + w] ^ ((i == encX) ? p1[w] : 0ull);
        /* parity block b goes to code bits [K + b z, K + (b + 1) z): assembled in words, then ORed in bytewise */
        const long long nbits = (long long)ldpcM;
        std::vector<unsigned long long> out(((size_t)ldpcN + 63) / 64 + 2, 0ull);
        for (int b = 0; b < mb; ++b) xor_bits(out.data(), (long long)ldpcK + (long long)b * z, &par[(size_t)b * W], 0, z);
        const unsigned char *ob = reinterpret_cast<const unsigned char *>(out.data());
        for (long long byte = ldpcK / 8; byte < (ldpcK + nbits + 7) / 8 && byte < ldpcN / 8; ++byte) code[byte] |= (char)ob[byte];
        return LDPC_SUCCESS;
    }
    std::vector<unsigned char> s((size_t)ldpcK, 0), par((size_t)ldpcM, 0);
    for (int c = 0; c < ldpcK / 8 && c < srcLength; ++c)         /* :641-650 */
        for (int b = 0; b < 8; ++b) s[(size_t)8 * c + b] = ((unsigned char)src[c] >> b) & 1;
    /* lambda_i = (A s) restricted to block row i */
    std::vector<unsigned char> lam((size_t)ldpcM, 0);
    for (int i = 0; i < mb; ++i)
        for (int j = 0; j < kb; ++j) {
            const int p = shift[(size_t)i * kSeedCols + j];
            if (p < 0) continue;
            for (int r = 0; r < z; ++r) lam[(size_t)i * z + r] ^= s[(size_t)j * z + (r + p) % z];
        }
    for (int i = 0; i < ldpcM; ++i) {
        unsigned char acc = 0;
        for (int j = 0; j < ldpcM; ++j) acc ^= denseInv[(size_t)i * ldpcM + j] & lam[j];
        par[i] = acc;
    }
    for (int i = 0; i < ldpcM; ++i)
        if (par[i]) {
            const int off = ldpcK + i;
            code[off / 8] |= (char)(1 << (off % 8));
        }
    return LDPC_SUCCESS;
}

/* ----------------------------------------------------------------- decoder */

/* forDecoder, MyLdpc.cpp:167-305: the adjacency lists become an ldpc_graph; the
 * OpenCL context/program/buffers become decoder handles made in addDecodeType. */
int Coder::forDecoder(int batchSize)
{
    if (!hSeed) return fail(LDPC_ERR_STATE, "Coder was not constructed");
    if (batchSize <= 0) return fail(LDPC_ERR_ARG, "forDecoder: batchSize must be positive");
    this->batchSize = batchSize;
    if (!graph) {
        int rc = ldpc_graph_create(rows.data(), cols.data(), nonZeros, ldpcM, ldpcN, &graph);
        if (rc) return fail(rc, ldpc_last_error());
    }
    isDecoder = true;
    return LDPC_SUCCESS;
}

/* one device (setDevice) or a device list (setDevices) behind the handle */
int Coder::makeDecoder(const ldpc_decoder_config &cfg0, ldpc_decoder **out)
{
    ldpc_decoder_config cfg = cfg0;
    cfg.host_input = hostInput;
    if (devices.empty()) return ldpc_decoder_create(graph, &cfg, out);
    return ldpc_decoder_create_multi(graph, &cfg, devices.data(), (int)devices.size(), out);
}

/* addDecodeType, MyLdpc.cpp:307-552. */
int Coder::addDecodeType(enum decodeType deType)
{
    if (!isDecoder) return fail(LDPC_ERR_STATE, "addDecodeType: call forDecoder() first");
    if (decoders.count((int)deType)) return LDPC_SUCCESS;
    if (deType == DecodeCPU) return LDPC_SUCCESS;   /* made on first use: needs the stream length (:438-439 is a no-op too) */
    ldpc_decoder_config cfg;
    ldpc_decoder_config_init(&cfg);
    cfg.K = ldpcK;
    cfg.max_batch = batchSize;
    cfg.max_iter = times;
    cfg.llr_scale = llrScale;
    cfg.device = device;
    cfg.early_term = 1;
    cfg.poll_interval = 4;
    cfg.pack_mode = LDPC_PACK_BYTES;
    cfg.layer_rows = z;
    cfg.algo = (deType == DecodeSP) ? LDPC_ALGO_SP : (deType == DecodeMS) ? LDPC_ALGO_MS
               : (deType == DecodeMSCL) ? LDPC_ALGO_MS_FUSED : LDPC_ALGO_LAYERED;
    if (deType == DecodeTDMP) {
        /* The reference's host-layered path (MyLdpc.cpp:889-976) sizes layer l as
         * hRowRange[l + z] - hRowRange[l] (:907, :958): right only when all rows of H have one weight
         * (rates 2/3A and 5/6).  There DecodeTDMP follows it operation for operation
         * (LDPC_ALGO_LAYERED_HOST); for the other seeds the reference's result is not a decode of H,
         * and DecodeTDMP runs the layered schedule with the fused kernel's semantics instead. */
        bool uniform = true;
        for (int m = 1; m < ldpcM && uniform; ++m)
            uniform = (rowRange[m + 1] - rowRange[m]) == (rowRange[1] - rowRange[0]);
        if (uniform) cfg.algo = LDPC_ALGO_LAYERED_HOST;
    }
    if (deType == DecodeMSCL) cfg.max_iter = 120;      /* hard-coded in the reference kernel, decodeCL.c:479 */
    ldpc_decoder *d = nullptr;
    int rc = makeDecoder(cfg, &d);
    if (rc) return fail(rc, ldpc_last_error());
    decoders[(int)deType] = d;
    return LDPC_SUCCESS;
}

/* decode, MyLdpc.cpp:571-618: the frame stream is cut into batchSize groups by
 * ldpc_decode.  srcLength bytes are written. */
int Coder::decode(float *postCode, char *srcCode, int srcLength, enum decodeType deType)
{
    if (!isDecoder) return fail(LDPC_ERR_STATE, "decode: call forDecoder() first");
    if (!postCode || !srcCode || srcLength <= 0) return fail(LDPC_ERR_ARG, "decode: bad arguments");
    const int codeSize = getCodeSize(srcLength);
    ldpc_decoder *d = nullptr;
    if (deType == DecodeCPU) {
        /* decodeCPU, MyLdpc.cpp:684-784: one pass over the whole stream, bit-offset packing */
        if (!decoders.count((int)DecodeCPU) || cpuDecoderBatch < codeSize) {
            if (decoders.count((int)DecodeCPU)) ldpc_decoder_destroy(decoders[(int)DecodeCPU]);
            decoders.erase((int)DecodeCPU);
            ldpc_decoder_config cfg;
            ldpc_decoder_config_init(&cfg);
            cfg.K = ldpcK; cfg.max_batch = codeSize; cfg.max_iter = times; cfg.device = device;
            cfg.algo = LDPC_ALGO_MS; cfg.pack_mode = LDPC_PACK_BITS; cfg.early_term = 1; cfg.poll_interval = 4;
            cfg.layer_rows = z;                              /* circulant size: lets the one-launch kernels apply */
            int rc = makeDecoder(cfg, &d);
            if (rc) return fail(rc, ldpc_last_error());
            decoders[(int)DecodeCPU] = d;
            cpuDecoderBatch = codeSize;
        }
        d = decoders[(int)DecodeCPU];
        memset(srcCode, 0, (size_t)srcLength);               /* :685 */
    } else {
        auto it = decoders.find((int)deType);
        if (it == decoders.end()) return fail(LDPC_ERR_STATE, "decode: addDecodeType() was not called for this type");
        d = it->second;
    }
    int rc = ldpc_decode(d, postCode, codeSize, (uint8_t *)srcCode, srcLength, nullptr);
    if (rc) return fail(rc, ldpc_last_error());
    ldpc_decode_stats st;
    if (ldpc_decoder_stats(d, &st) == LDPC_OK) lastTime = st.batch_time;
    return LDPC_SUCCESS;
}

/* ------------------------------------------------------------ test channel */

/* gaussian, MyLdpc.cpp:1093-1105.  C++ overloads: sqrt/log/cos on float. */
float gaussian(float ave, float sd)
{
    const float pi = 3.1415926f;
    const float s1 = (float)((1.0 + rand()) / (RAND_MAX + 1.0));
    const float s2 = (float)((1.0 + rand()) / (RAND_MAX + 1.0));
    const float r = std::sqrt(-2 * std::log(s2));
    const float t = 2 * pi * s1;
    const float zz = r * std::cos(t);
    return ave + zz * sd;
}

/* test, MyLdpc.cpp:1061-1078 */
int Coder::test(char *priorCode, float *postCode, int priorCodeLength, float rate)
{
    if (!priorCode || !postCode || priorCodeLength < 0) return fail(LDPC_ERR_ARG, "test: bad arguments");
    for (int c = 0; c < priorCodeLength; ++c)
        for (int b = 0; b < 8; ++b)
            postCode[c * 8 + b] = (priorCode[c] & (1 << b)) ? -1.0f : 1.0f;
    for (int i = 0; i < priorCodeLength * 8; ++i) postCode[i] += gaussian(0, rate);
    return LDPC_SUCCESS;
}
