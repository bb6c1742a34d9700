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
    // LDS-tiled kernel: work descriptors int32[ndesc][8] = {q_begin, q_end, win_begin, win_end, slab, 0, 0, 0}
    // over the row records rowinfo int32[m][12] (layout: include/sparc_hip.h)
    const int32_t* desc;
    const int32_t* rowinfo;
    bool rec_mirrored;  // single-width tiled pass: `val` holds the mirrored values, so take valT of the records' outside entries
    int dbg;  // ablation switches for kernel timing experiments (0 in production)
    int ndesc;
    int tile_rows;  // largest window (rows) over all descriptors
    int kp;         // columns per half (dual) or per row (single) of the global operand
    int wtot;       // floats per row of the global operand
    // affinity preparation epilogue
    double* racc;  // [m][8] {sum sq, S 1, S c, S c2 (all off-diagonal), S_ii}
    const double* cvec;
    const double* c2vec;
    int64_t col0;
};

// LDS budget of one tile: SPARC_TILE_MAX_ROWS rows x 256 bytes
#define SPARC_TILE_LDS_BYTES (SPARC_TILE_MAX_ROWS * 256)

int sparc_spmm_launch(int epi, int width, const SpmmArgs& a, hipStream_t stream);
int64_t sparc_rows_blocks(int64_t m);
