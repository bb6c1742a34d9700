// Internal interface of the row-gather product kernels (spmm.hip), shared with the
// affinity preparation (affinity_prep.hip).
#pragma once
#include "common.h"

enum SpmmEpilogue {
    EPI_STORE = 0,       // out = A Xg
    EPI_ADD_X0 = 1,      // out = X0 + A Xg
    EPI_ADD_IDENT = 2,   // out = E + A Xg, E = columns [col0, col0 + W) of the identity
    EPI_LAP_SINGLE = 3,  // last pass, loss only (validation)
    EPI_LAP_DUAL = 4,    // last pass, loss + gradient
    EPI_PREP_ACC = 5,    // row statistics of S[:, col0 : col0 + W)
};

// The column-sum finish of the Laplacian chain (spmm.hip: colsum_finish_body), which the first inner pass can
// carry spread over its workgroups; colpart == null: none.
// most workgroups of lap_scale = most block partials of the chain's column sums (a multiple of 64)
#ifndef SPARC_SCALE_MAX_BLOCKS
#define SPARC_SCALE_MAX_BLOCKS 128
#endif
struct ColsumFinish {
    const float* colpart;    // [nparts][6][kp] block partials of lap_scale
    const double* coupling;  // the 3 x 3 coupling
    float* M12;              // [6][kp] + 9
    int kp, nparts, dual;
};

struct SpmmArgs {
    int64_t m;
    int64_t nnz;
    const int64_t* rowptr;
    const int32_t* col;
    const float* val;
    const float* valT;
    const float* Xg;  // gathered operand  [m][W]
    const float* X0;  // own-row operand   [m][W]
    float* out;       // [m][W]
    const float* addend;  // row kernel: [m][W] added to the row's product before the epilogue (the dense-block part), or null
    int skip_entries;     // row kernel: the addend is the whole product, no entry is gathered (epilogue only)
    int interleave_rows;  // row kernel, plain products: workgroup b takes rows 4 b .. 4 b + 3 (consecutive workgroups run on different XCDs)
    int64_t split_above;  // row kernel, plain products: rows of more entries are left to spmm_chunks / spmm_merge (0: none)
    // Laplacian epilogues
    const float* rowc;  // [m][SPARC_ROWC]
    const float* M12;   // [6][KP]
    double* loss_part;  // one per block
    double* loss_out;   // the last workgroup to arrive writes loss_scale * (sum of loss_part) here
    double loss_scale;
    int chain_slot;     // which set of arrival counters this call uses
    int rows_finish;    // row kernel, loss epilogue: the last workgroup to arrive sums the partials (no loss_finish launch)
    ColsumFinish fin;   // EPI_ADD_X0: workgroup c < fin.kp also finishes column c of the column sums
    float* gradY;
    int64_t ldy;
    int k;
    float inv_m;
    // windowed kernel: work descriptors int32[ndesc][8], slot records int32[.][4] and the entry streams
    // (layout: include/sparc_hip.h)
    const int32_t* desc;
    const int32_t* rowinfo;
    const int32_t* s_addr;
    const float* s_val;
    const float* s_valT;
    const int32_t* o_row;
    const float* o_val;
    const float* o_valT;
    const float* rowpat;  // pattern mode of the windowed kernel: float[m][4] = {r, delta, delta / r, 1 / r}, else null
    int ndesc;
    int tile_rows;   // largest window (rows) over all descriptors
    int vec;         // floats per lane: a slab row is 16 * vec floats
    int threads;     // workgroup size
    int64_t plane;   // floats between two slabs of the slab-major operand (m * 16 * vec)
    int kp;          // columns per half (dual) or per row (single): the stride of M12
    long long* stamps;  // developer timeline buffer or null
    int dbg;         // ablation switches for kernel timing experiments (0 in production)
    // affinity preparation epilogue
    double* racc;  // [m][8] {sum sq, S 1, S c, S c2 (all off-diagonal), S_ii}
    const double* cvec;
    const double* c2vec;
    int64_t col0;
    const uint8_t* gflag;  // [m]: nonzero where the row of Xg holds anything (the other rows are not read)
    uint8_t* oflag;        // [m]: EPI_ADD_IDENT writes it for out, and stores only the rows that hold anything
};

// LDS budget of one window
#define SPARC_TILE_LDS_BYTES (128 * 1024)

// Dense diagonal blocks (blocks.hip): out = blockdiag(D_c) Xg (+ X0) on row-major [m][W] operands
struct BlocksArgs {
    const int32_t* desc;  // int32[ndesc][8]: {first row of the cluster, cluster rows, P, stripe, offset low 31 bits, offset high bits, 0, 0}
    int ndesc;
    int max_p;            // largest P
    const float* blk;     // the blocks, stripe by stripe in k-major order (layout: include/sparc_hip.h)
    const float* blkT;    // blocks of the mirrored values (== blk for a symmetric matrix or a forward-only pass)
    const float* Xg;
    const float* X0;
    float* out;
    int W;                // operand width, a multiple of 32 up to 128
    int split;            // first column of the A^T half (W when there is none)
    // fused launch: the entries outside the blocks, CSR (r_valT == r_val for symmetric values)
    const int64_t* r_rowptr;
    const int32_t* r_col;
    const float* r_val;
    const float* r_valT;
    int dbg;              // ablation switches for kernel timing experiments (0 in production)
    // fused launch of a training pass's LAST product: the loss / gradient epilogue on the finished rows (lap_rowc null: none)
    const float* lap_rowc;      // float[m][SPARC_ROWC]
    const float* lap_M12;       // [M1; M2; coupling], rows of lap_kp floats
    int lap_kp, lap_k;          // columns per half (W == 2 lap_kp), columns of Y
    float lap_inv_m;
    float* lap_gradY;           // [m][lap_ldy]
    int64_t lap_ldy;
    double* lap_loss_part;      // one partial per workgroup
    double* lap_loss_out;       // the last workgroup to arrive writes lap_loss_scale * (their sum) here
    double lap_loss_scale;
    unsigned int* lap_counter;  // this call's arrival counters (device memory; chain.h)
    // fused launch of a training pass's FIRST product: the finish of the chain's column sums rides along (colpart null: none)
    ColsumFinish fin;
};
#define SPARC_BLOCKS_LDS_BYTES (152 * 1024)
size_t sparc_blocks_lds_bytes(int max_p, int W, bool mirrored, bool fused, bool lap = false);
bool sparc_blocks_can_fuse(int max_p, int W, bool mirrored);
int sparc_blocks_launch(const BlocksArgs& a, bool add, bool fused, hipStream_t stream);

int sparc_spmm_launch(int epi, int width, const SpmmArgs& a, hipStream_t stream);
int64_t sparc_rows_blocks(int64_t m);
