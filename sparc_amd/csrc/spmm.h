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
    // Laplacian epilogues
    const float* rowc;  // [m][SPARC_ROWC]
    const float* M12;   // [6][KP]
    double* loss_part;  // one per block
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
};

// LDS budget of one window
#define SPARC_TILE_LDS_BYTES (128 * 1024)

int sparc_spmm_launch(int epi, int width, const SpmmArgs& a, hipStream_t stream);
int64_t sparc_rows_blocks(int64_t m);
