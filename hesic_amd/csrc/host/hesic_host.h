/* hesic_host.h -- C ABI of libhesic_host.so: the host-side (CPU, sequential) entropy-coding helpers that
 * the reference ships as two pybind11 extensions and imports at module load time:
 *   compressai._CXX.pmf_to_quantized_cdf   (compressai/cpp_exts/ops/ops.cpp:24-81)
 *   compressai.ans.{BufferedRansEncoder,RansEncoder,RansDecoder} (compressai/cpp_exts/rans/rans_interface.cpp)
 * Byte-exact with the reference coder: 64-bit rANS state, 32-bit word emission, lower bound 2^31,
 * 16-bit probabilities, 4-bit bypass escape for symbols outside the CDF support.
 * All pointers are HOST pointers.  Return value 0 = ok, -1 = bad argument.                              */
#ifndef HESIC_HOST_H
#define HESIC_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* cdf_out has n+1 entries; cdf_out[0]=0, cdf_out[n]=1<<precision, strictly increasing. */
int hesic_pmf_to_quantized_cdf(const float* pmf, int n, int precision, uint32_t* cdf_out);

/* The same for every row of a [rows][pmf_stride] matrix in one call (EntropyBottleneck.update / GaussianConditional.update,
 * compressai/entropy_models/entropy_models.py:136-142): row r = pmf[r][0..lengths[r]) + its tail-mass escape bin;
 * cdf_out [rows][cdf_stride] int32 gets lengths[r]+2 entries per row, zero padded.                       */
int hesic_pmf_rows_to_quantized_cdfs(const float* pmf, int rows, int pmf_stride, const int32_t* lengths, const float* tail_mass,
                                     int precision, int32_t* cdf_out, int cdf_stride);

typedef struct hesic_rans_encoder hesic_rans_encoder;
typedef struct hesic_rans_decoder hesic_rans_decoder;

hesic_rans_encoder* hesic_rans_encoder_new(void);
void hesic_rans_encoder_free(hesic_rans_encoder*);
/* Queue n symbols.  cdfs: [ncdf][cdf_stride] int32 row-major; cdf_sizes/offsets: [ncdf]. */
int hesic_rans_encoder_push(hesic_rans_encoder*, const int32_t* symbols, const int32_t* indexes, int64_t n,
                            const int32_t* cdfs, int ncdf, int cdf_stride, const int32_t* cdf_sizes,
                            const int32_t* offsets);
/* Encode everything queued (last symbol first) and reset.  Returns the stream size in bytes; copies it to
 * out if cap is large enough (call with out=NULL to size the buffer: the queue is kept in that case).   */
int64_t hesic_rans_encoder_flush(hesic_rans_encoder*, uint8_t* out, int64_t cap);

hesic_rans_decoder* hesic_rans_decoder_new(void);
void hesic_rans_decoder_free(hesic_rans_decoder*);
int hesic_rans_decoder_set_stream(hesic_rans_decoder*, const uint8_t* bytes, int64_t nbytes);
int hesic_rans_decoder_decode(hesic_rans_decoder*, const int32_t* indexes, int64_t n, const int32_t* cdfs, int ncdf,
                              int cdf_stride, const int32_t* cdf_sizes, const int32_t* offsets, int32_t* symbols_out);

/* ---- adaptive range coder of HSIC.compress / decompress (ywz/mywork/newnet1.py:905-1040, :1137-1240): every symbol
 * comes with its own cumulative-frequency table (cdf[0] = 0 ... cdf[A] = total, total need not be a power of two), the
 * two views share one stream and the decoder is fed view 2's tables only after view 1 has been decoded.  The reference
 * delegates this to the third-party `range_coder` package (PyPI, unpinned, absent from the reference tree): this is the
 * carry-less range coder of that family (64-bit low, 2^56 top / 2^48 bottom, byte-wise renormalisation), restated from
 * the published algorithm -- round trips are exact, byte-compatibility with `range_coder` is NOT pinned.
 * cdf: [n][stride] uint32 row-major, row i = table of symbol i (stride = alphabet + 1).                               */
typedef struct hesic_rc_encoder hesic_rc_encoder;
typedef struct hesic_rc_decoder hesic_rc_decoder;
hesic_rc_encoder* hesic_rc_encoder_new(void);
void hesic_rc_encoder_free(hesic_rc_encoder*);
/* -1: bad argument, -2: a symbol with zero frequency / outside its table */
int hesic_rc_encoder_encode(hesic_rc_encoder*, const int32_t* symbols, const uint32_t* cdf, int64_t n, int32_t stride);
/* finishes the stream (idempotent); returns its size and copies it to out when cap suffices (out=NULL: size only) */
int64_t hesic_rc_encoder_finish(hesic_rc_encoder*, uint8_t* out, int64_t cap);
hesic_rc_decoder* hesic_rc_decoder_new(const uint8_t* bytes, int64_t nbytes);     /* copies the stream */
void hesic_rc_decoder_free(hesic_rc_decoder*);
int hesic_rc_decoder_decode(hesic_rc_decoder*, const uint32_t* cdf, int64_t n, int32_t stride, int32_t* symbols_out);
/* symbol (p, q), p outer, under table row p * row_step_outer + q * row_step_inner (tables in another order than the stream) */
int hesic_rc_decoder_decode_grid(hesic_rc_decoder*, const uint32_t* cdf, int64_t n_outer, int64_t n_inner, int64_t row_step_outer,
                                 int64_t row_step_inner, int32_t stride, int32_t* symbols_out);

#ifdef __cplusplus
}
#endif
#endif
