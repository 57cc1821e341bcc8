// Device-side construction of the assembly plan (pattern of R' H_blk R + contribution lists); see plan_device.hip.
#pragma once
#include <cstdint>

#include "common.hpp"
#include "problem.hpp"

namespace mgbhip {

struct PlanDeviceIn {
    bool selection;
    int64_t NE, n, m;
    int32_t p, nu;
    // general levels: per-(element, state) column sets and the offsets of the projected slab blocks (device)
    const int32_t* ecol_ptr;
    const int32_t* ecols;
    const int32_t* eoff;
    int64_t slab_doubles;
    // selection levels: R in CSR (device), identity-only states, compact diagonal blocks of the element slab
    const int32_t* Rptr;
    const int32_t* Rcol;
    uint32_t state_id_mask;
    int32_t diag_mask_sel;
    int64_t sel_off[MGBHIP_MAX_NU * (MGBHIP_MAX_NU + 1) / 2];
    // > 0: also build the direct-value map (Level::h_vmap, sh_q, nshared); slots of shared entries start here
    int64_t extra_base;
};

// above this the transient sort buffers (about 40 B per pair) are not worth it: the host builder takes over
constexpr int64_t PLAN_DEVICE_MAX_PAIRS = 1ll << 28;

// number of (key, source) pairs the device builder sorts for this level
int64_t plan_device_pairs(const PlanDeviceIn& in);
// fills L.Hptr/Hcol/cptr/cidx (device), L.hHptr/hHcol (host copy for the symbolic analysis), L.nnz, L.long_lists
void build_plan_device(const PlanDeviceIn& in, Level& L, hipStream_t st);

// CSR of R' (the gather form of R' * v) from the CSR of R on the device: one stable radix sort of the entries by column,
// so the entries of a row of R' come in ascending row order of R -- the order of the host loop this replaces.
// Returns the longest row of R' (what decides the long-row / chunked restriction kernels).
int32_t transpose_csr_device(int64_t rows, int64_t cols, int64_t nnz, const int32_t* Rptr, const int32_t* Rcol, const double* Rval,
                             DevBuf<int32_t>& Tptr, DevBuf<int32_t>& Tcol, DevBuf<double>& Tval, hipStream_t st);

}  // namespace mgbhip
