// Assembly-plan construction on the device.
//
// reference: `BlockAssemblyPlan` construction, src/BlockMatrices.jl:322-491 (host dictionaries keyed by
// (row, col) that map every structural nonzero of R' * H_blk * R to its element contributions).  Here the
// same map is one stable radix sort: every element contribution becomes a ((row, col) key, slab index)
// pair in element order, rocPRIM sorts the pairs by key, run heads give the CSR pattern of H and the run
// bodies are the contribution lists.  The sort is stable, so each list keeps element order and the
// deterministic summation order of the gather kernel is the one the host plan builder produces.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "plan_device.hpp"

namespace mgbhip {

namespace {

constexpr uint32_t SENTINEL = 0xFFFFFFFFu;     // "keep a diagonal slot" pair: key only, no contribution

// general level: the contributions of element e are the ct x ct entries of its projected slab block
__global__ void pairs_general(int64_t NE, int nu, int64_t m, const int32_t* __restrict__ ecol_ptr,
                              const int32_t* __restrict__ ecols, const int32_t* __restrict__ eoff,
                              uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    for (int64_t e = blockIdx.x; e < NE; e += gridDim.x) {
        const int32_t base = ecol_ptr[e * nu];
        const int32_t ct = ecol_ptr[(e + 1) * nu] - base;
        const int32_t o = eoff[e];
        for (int32_t q = threadIdx.x; q < ct * ct; q += blockDim.x) {
            const int32_t gi = q % ct, gj = q / ct;
            keys[o + q] = (uint64_t)ecols[base + gi] * (uint64_t)m + (uint64_t)ecols[base + gj];
            vals[o + q] = (uint32_t)(o + q);
        }
    }
}

// selection level: slot ((a*p + i)*nu + b)*p + j of element e; rows of R hold at most one entry
__global__ void pairs_selection(PlanDeviceIn in, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                int32_t* __restrict__ err) {
    const int64_t slots = (int64_t)in.nu * in.p * in.nu * in.p;
    const int64_t total = in.NE * slots;
    const uint64_t invalid = (uint64_t)in.m * (uint64_t)in.m;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = t / slots;
        int r = (int)(t - e * slots);
        const int j = r % in.p; r /= in.p;
        const int b = r % in.nu; r /= in.nu;
        const int i = r % in.p;
        const int a = r / in.p;
        uint64_t key = invalid;
        uint32_t val = 0;
        const bool structural = !(((in.state_id_mask >> a) & 1) && ((in.state_id_mask >> b) & 1) && i != j);
        if (structural) {
            const int64_t ra = (int64_t)a * in.n + e * in.p + i, rb = (int64_t)b * in.n + e * in.p + j;
            const int32_t pa = in.Rptr[ra], pb = in.Rptr[rb];
            if (in.Rptr[ra + 1] > pa && in.Rptr[rb + 1] > pb) {
                const int32_t ci = in.Rcol[pa], cj = in.Rcol[pb];
                const int lo = a <= b ? a : b, hi = a <= b ? b : a;
                const int blk = lo * in.nu - lo * (lo - 1) / 2 + (hi - lo);
                int64_t src;
                if ((in.diag_mask_sel >> blk) & 1) src = in.sel_off[blk] + e * in.p + i;
                else if (a <= b) src = in.sel_off[blk] + (e * in.p + j) * (int64_t)in.p + i;
                else src = in.sel_off[blk] + (e * in.p + i) * (int64_t)in.p + j;
                if (src >= (int64_t)INT32_MAX) *err = 1;
                key = (uint64_t)ci * (uint64_t)in.m + (uint64_t)cj;
                val = (uint32_t)src;
            }
        }
        keys[t] = key;
        vals[t] = val;
    }
}

__global__ void pairs_diagonal(int64_t m, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) {
        keys[i] = (uint64_t)i * (uint64_t)m + (uint64_t)i;
        vals[i] = SENTINEL;
    }
}

__device__ inline int64_t lower_bound_u64(const uint64_t* __restrict__ k, int64_t n, uint64_t x) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (k[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ void count_valid(const uint64_t* __restrict__ keys, int64_t P, uint64_t invalid, int64_t* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = lower_bound_u64(keys, P, invalid);
}

// packed flags: high word = run head (a new structural nonzero), low word = a real contribution
__global__ void run_flags(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, int64_t nvalid,
                          uint64_t* __restrict__ flags) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nvalid; t += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t head = (t == 0 || keys[t] != keys[t - 1]) ? 1ull : 0ull;
        const uint64_t con = vals[t] != SENTINEL ? 1ull : 0ull;
        flags[t] = (head << 32) | con;
    }
}

__global__ void scatter_plan(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                             const uint64_t* __restrict__ flags, const uint64_t* __restrict__ scan, int64_t nvalid,
                             int64_t m, int32_t* __restrict__ Hcol, int32_t* __restrict__ cptr, int32_t* __restrict__ cidx) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nvalid; t += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t f = flags[t], s = scan[t];
        const uint32_t q = (uint32_t)(s >> 32), cp = (uint32_t)(s & 0xFFFFFFFFull);
        if (f >> 32) {
            Hcol[q] = (int32_t)(keys[t] % (uint64_t)m);
            cptr[q] = (int32_t)cp;
        }
        if (f & 1ull) cidx[cp] = (int32_t)vals[t];
    }
}

__global__ void row_pointers(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ scan, int64_t nvalid,
                             int64_t m, int32_t nnz, int32_t total, int32_t* __restrict__ Hptr, int32_t* __restrict__ cptr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= m) {
        const int64_t lb = lower_bound_u64(keys, nvalid, (uint64_t)i * (uint64_t)m);
        Hptr[i] = lb < nvalid ? (int32_t)(scan[lb] >> 32) : nnz;
    }
    if (i == 0) cptr[nnz] = total;
}

// Direct-value map of a selection level: a structural nonzero with exactly one contribution IS an entry of
// the element-block slab; the others (shared between elements, or empty) get a slot in a compact "shared" array
// behind the slab.  vmap[q] = slab index, or extra_base + rank among the shared ones.
__global__ void shared_flags(const int32_t* __restrict__ cptr, int64_t nnz, uint32_t* __restrict__ flag) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nnz; q += (int64_t)gridDim.x * blockDim.x)
        flag[q] = (cptr[q + 1] - cptr[q] != 1) ? 1u : 0u;
}

__global__ void value_map(const int32_t* __restrict__ cptr, const int32_t* __restrict__ cidx, const uint32_t* __restrict__ flag,
                          const uint32_t* __restrict__ rank, int64_t nnz, int32_t extra_base, int32_t* __restrict__ vmap,
                          int32_t* __restrict__ sh_q) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nnz; q += (int64_t)gridDim.x * blockDim.x) {
        if (flag[q]) {
            vmap[q] = extra_base + (int32_t)rank[q];
            sh_q[rank[q]] = (int32_t)q;
        } else {
            vmap[q] = cidx[cptr[q]];
        }
    }
}

struct PlusU32 {
    __host__ __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a + b; }
};

struct PlusU64 {
    __host__ __device__ uint64_t operator()(uint64_t a, uint64_t b) const { return a + b; }
};

}  // namespace

int64_t plan_device_pairs(const PlanDeviceIn& in) {
    const int64_t body = in.selection ? in.NE * (int64_t)in.nu * in.p * in.nu * in.p : in.slab_doubles;
    return body + in.m;
}

void build_plan_device(const PlanDeviceIn& in, Level& L, hipStream_t st) {
    const int64_t m = in.m;
    const int64_t P = plan_device_pairs(in);
    const int64_t body = P - m;
    MGB_REQUIRE(m > 0 && P < (int64_t)INT32_MAX, "device plan: pair count exceeds 32-bit indexing");
    DevBuf<uint64_t> k0, k1, flags, scan;
    DevBuf<uint32_t> v0, v1;
    DevBuf<int64_t> d_nvalid;
    DevBuf<int32_t> d_err;
    k0.alloc((size_t)P); k1.alloc((size_t)P); v0.alloc((size_t)P); v1.alloc((size_t)P);
    d_nvalid.alloc(1); d_err.alloc(1);
    d_err.zero(st);
    if (body > 0) {
        if (in.selection) {
            const int64_t blocks = std::min<int64_t>((body + 255) / 256, 1 << 16);
            hipLaunchKernelGGL(pairs_selection, dim3((unsigned)blocks), dim3(256), 0, st, in, k0.p, v0.p, d_err.p);
        } else {
            const int64_t blocks = std::min<int64_t>(in.NE, 1 << 16);
            hipLaunchKernelGGL(pairs_general, dim3((unsigned)blocks), dim3(256), 0, st, in.NE, in.nu, m, in.ecol_ptr, in.ecols,
                               in.eoff, k0.p, v0.p);
        }
    }
    hipLaunchKernelGGL(pairs_diagonal, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, m, k0.p + body, v0.p + body);
    MGB_HIP_CHECK(hipGetLastError());
    // stable sort by (row, col): keys are < m*m (+1 for the invalid marker)
    unsigned bits = 1;
    while (bits < 64 && (((uint64_t)m * (uint64_t)m) >> bits) != 0) ++bits;
    size_t tmp_bytes = 0;
    MGB_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0.p, k1.p, v0.p, v1.p, (size_t)P, 0u, bits, st));
    DevBuf<char> tmp;
    tmp.alloc(tmp_bytes + 16);
    MGB_HIP_CHECK(rocprim::radix_sort_pairs((void*)tmp.p, tmp_bytes, k0.p, k1.p, v0.p, v1.p, (size_t)P, 0u, bits, st));
    hipLaunchKernelGGL(count_valid, dim3(1), dim3(64), 0, st, k1.p, P, (uint64_t)m * (uint64_t)m, d_nvalid.p);
    int64_t nvalid = 0;
    int32_t err = 0;
    MGB_HIP_CHECK(hipMemcpyAsync(&nvalid, d_nvalid.p, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    MGB_HIP_CHECK(hipMemcpyAsync(&err, d_err.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    MGB_HIP_CHECK(hipStreamSynchronize(st));
    MGB_REQUIRE(err == 0, "slab exceeds 32-bit indexing");
    MGB_REQUIRE(nvalid >= m, "device plan: lost the diagonal pairs");
    // k0 / v0 are free again: reuse k0 as the flag array
    flags = std::move(k0);
    scan.alloc((size_t)nvalid);
    const int64_t blocks = std::min<int64_t>((nvalid + 255) / 256, 1 << 16);
    hipLaunchKernelGGL(run_flags, dim3((unsigned)blocks), dim3(256), 0, st, k1.p, v1.p, nvalid, flags.p);
    size_t scan_bytes = 0;
    MGB_HIP_CHECK(rocprim::exclusive_scan(nullptr, scan_bytes, flags.p, scan.p, (uint64_t)0, (size_t)nvalid, PlusU64(), st));
    tmp.ensure(scan_bytes + 16);
    MGB_HIP_CHECK(rocprim::exclusive_scan((void*)tmp.p, scan_bytes, flags.p, scan.p, (uint64_t)0, (size_t)nvalid, PlusU64(), st));
    uint64_t last_scan = 0, last_flag = 0;
    MGB_HIP_CHECK(hipMemcpyAsync(&last_scan, scan.p + (nvalid - 1), sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    MGB_HIP_CHECK(hipMemcpyAsync(&last_flag, flags.p + (nvalid - 1), sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    MGB_HIP_CHECK(hipStreamSynchronize(st));
    const uint64_t tot = last_scan + last_flag;
    const int64_t nnz = (int64_t)(tot >> 32), total = (int64_t)(tot & 0xFFFFFFFFull);
    MGB_REQUIRE(nnz < (int64_t)INT32_MAX, "Hessian pattern exceeds 32-bit indexing");
    MGB_REQUIRE(total < (int64_t)INT32_MAX, "contribution list exceeds 32-bit indexing");
    L.nnz = nnz;
    L.Hptr.alloc((size_t)m + 1);
    L.Hcol.alloc((size_t)nnz);
    L.cptr.alloc((size_t)nnz + 1);
    L.cidx.alloc((size_t)std::max<int64_t>(total, 1));
    hipLaunchKernelGGL(scatter_plan, dim3((unsigned)blocks), dim3(256), 0, st, k1.p, v1.p, flags.p, scan.p, nvalid, m, L.Hcol.p,
                       L.cptr.p, L.cidx.p);
    hipLaunchKernelGGL(row_pointers, dim3((unsigned)((m + 1 + 255) / 256)), dim3(256), 0, st, k1.p, scan.p, nvalid, m, (int32_t)nnz,
                       (int32_t)total, L.Hptr.p, L.cptr.p);
    MGB_HIP_CHECK(hipGetLastError());
    // the symbolic analysis of the direct solver runs on the host: it needs the pattern
    L.hHptr.resize((size_t)m + 1);
    L.hHcol.resize((size_t)nnz);
    L.Hptr.download(L.hHptr.data(), (size_t)m + 1, st);
    L.Hcol.download(L.hHcol.data(), (size_t)nnz, st);
    MGB_HIP_CHECK(hipStreamSynchronize(st));
    L.long_lists = nnz > 0 && total / nnz > 48;
    // direct-value map (selection levels): lets the solver read single-contribution entries straight from the slab
    L.direct = false;
    if (in.selection && in.extra_base > 0 && nnz > 0 && in.extra_base + nnz + m + 1 < (int64_t)INT32_MAX) {
        DevBuf<uint32_t> fl, rk;
        DevBuf<int32_t> vmap;
        fl.alloc((size_t)nnz); rk.alloc((size_t)nnz); vmap.alloc((size_t)nnz);
        L.sh_q.alloc((size_t)nnz);          // upper bound; only the first nshared entries are used
        const int64_t qb = std::min<int64_t>((nnz + 255) / 256, 1 << 16);
        hipLaunchKernelGGL(shared_flags, dim3((unsigned)qb), dim3(256), 0, st, L.cptr.p, nnz, fl.p);
        size_t sb = 0;
        MGB_HIP_CHECK(rocprim::exclusive_scan(nullptr, sb, fl.p, rk.p, (uint32_t)0, (size_t)nnz, PlusU32(), st));
        tmp.ensure(sb + 16);
        MGB_HIP_CHECK(rocprim::exclusive_scan((void*)tmp.p, sb, fl.p, rk.p, (uint32_t)0, (size_t)nnz, PlusU32(), st));
        hipLaunchKernelGGL(value_map, dim3((unsigned)qb), dim3(256), 0, st, L.cptr.p, L.cidx.p, fl.p, rk.p, nnz, (int32_t)in.extra_base,
                           vmap.p, L.sh_q.p);
        uint32_t lr = 0, lf = 0;
        MGB_HIP_CHECK(hipMemcpyAsync(&lr, rk.p + (nnz - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        MGB_HIP_CHECK(hipMemcpyAsync(&lf, fl.p + (nnz - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        L.h_vmap.resize((size_t)nnz);
        vmap.download(L.h_vmap.data(), (size_t)nnz, st);
        MGB_HIP_CHECK(hipStreamSynchronize(st));
        L.nshared = (int64_t)lr + lf;
        if (L.nshared < nnz) {               // keep only the used part of the list
            DevBuf<int32_t> sq;
            sq.alloc((size_t)std::max<int64_t>(L.nshared, 1));
            if (L.nshared) MGB_HIP_CHECK(hipMemcpyAsync(sq.p, L.sh_q.p, (size_t)L.nshared * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
            MGB_HIP_CHECK(hipStreamSynchronize(st));
            L.sh_q = std::move(sq);
        }
        L.direct = true;
    }
}

namespace {

__global__ void expand_rows(int64_t rows, const int32_t* __restrict__ ptr, int32_t* __restrict__ rowidx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    for (int32_t q = ptr[i]; q < ptr[i + 1]; ++q) rowidx[q] = (int32_t)i;
}

__global__ void iota32(int64_t n, uint32_t* __restrict__ v) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}

// after the sort: entry d of R' is entry q = perm[d] of R; row pointers of R' from the sorted column keys
__global__ void transpose_fill(int64_t nnz, int64_t cols, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ perm,
                               const int32_t* __restrict__ rowidx, const double* __restrict__ Rval, int32_t* __restrict__ Tptr,
                               int32_t* __restrict__ Tcol, double* __restrict__ Tval) {
    const int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (d >= nnz) return;
    const uint32_t q = perm[d], c = keys[d];
    Tcol[d] = rowidx[q];
    Tval[d] = Rval[q];
    const uint32_t prev = d == 0 ? 0u : keys[d - 1];
    if (d == 0) for (uint32_t j = 0; j <= c; ++j) Tptr[j] = 0;
    else for (uint32_t j = prev + 1; j <= c; ++j) Tptr[j] = (int32_t)d;       // rows prev+1 .. c start here (empty ones included)
    if (d == nnz - 1) for (int64_t j = (int64_t)c + 1; j <= cols; ++j) Tptr[j] = (int32_t)nnz;
}

__global__ void max_row_length(int64_t rows, const int32_t* __restrict__ ptr, int32_t* __restrict__ out) {
    int32_t best = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows; i += (int64_t)gridDim.x * 256) best = max(best, ptr[i + 1] - ptr[i]);
    atomicMax(out, best);
}

}  // namespace

int32_t transpose_csr_device(int64_t rows, int64_t cols, int64_t nnz, const int32_t* Rptr, const int32_t* Rcol, const double* Rval,
                             DevBuf<int32_t>& Tptr, DevBuf<int32_t>& Tcol, DevBuf<double>& Tval, hipStream_t st) {
    Tptr.alloc((size_t)cols + 1);
    Tcol.alloc((size_t)std::max<int64_t>(nnz, 1));
    Tval.alloc((size_t)std::max<int64_t>(nnz, 1));
    if (nnz == 0) {
        Tptr.zero(st);
        MGB_HIP_CHECK(hipStreamSynchronize(st));
        return 0;
    }
    DevBuf<int32_t> rowidx, d_max;
    DevBuf<uint32_t> k1, v0, v1;
    rowidx.alloc((size_t)nnz); k1.alloc((size_t)nnz); v0.alloc((size_t)nnz); v1.alloc((size_t)nnz);
    d_max.alloc(1);
    d_max.zero(st, 1);
    hipLaunchKernelGGL(expand_rows, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, rows, Rptr, rowidx.p);
    hipLaunchKernelGGL(iota32, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, nnz, v0.p);
    unsigned bits = 1;
    while (bits < 32 && ((uint64_t)cols >> bits) != 0) ++bits;
    const uint32_t* k0 = reinterpret_cast<const uint32_t*>(Rcol);           // column indices are non-negative
    size_t tmp_bytes = 0;
    MGB_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1.p, v0.p, v1.p, (size_t)nnz, 0u, bits, st));
    DevBuf<char> tmp;
    tmp.alloc(tmp_bytes + 16);
    MGB_HIP_CHECK(rocprim::radix_sort_pairs((void*)tmp.p, tmp_bytes, k0, k1.p, v0.p, v1.p, (size_t)nnz, 0u, bits, st));
    hipLaunchKernelGGL(transpose_fill, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, nnz, cols, k1.p, v1.p, rowidx.p, Rval,
                       Tptr.p, Tcol.p, Tval.p);
    hipLaunchKernelGGL(max_row_length, dim3((unsigned)std::min<int64_t>((cols + 255) / 256, 1024)), dim3(256), 0, st, cols, Tptr.p, d_max.p);
    MGB_HIP_CHECK(hipGetLastError());
    int32_t maxrow = 0;
    d_max.download(&maxrow, 1, st);
    MGB_HIP_CHECK(hipStreamSynchronize(st));                                 // the sort buffers are locals
    return maxrow;
}

}  // namespace mgbhip
