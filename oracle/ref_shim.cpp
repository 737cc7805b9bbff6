// oracle/_ref/libref_shim.so -- a C-callable window onto the reference's OWN prebuilt native modules.
//
// TEST INFRASTRUCTURE ONLY (same rule as the rest of oracle/): used by tests/golden/make_golden*.py and
// by CPU tests that skip when it is absent.  It exists only in the build container: it links against
//   /root/reference/stgraph/graph/static/csr.so        (nvcc build of graph/static/csr.cu)
//   /root/reference/stgraph/graph/dynamic/pcsr/pcsr.so (nvcc build of graph/dynamic/pcsr/pcsr.cu)
// where they lie.  Both are pybind11 modules for CPython 3.8 (this image has 3.10, so `import` refuses
// them) with a statically linked CUDA runtime, but their C++ classes are ordinary exported symbols
// (`nm -C`: CSR::CSR(std::vector<std::tuple<int,int,int>>, std::vector<float>, int, bool),
// PCSR::edge_update_list(...), PCSR::build_csr(), ...).  This file declares the two classes with the
// member layout of the reference's declarations (csr.cu:35-59, pcsr.cu:273-318) -- a binding, the way
// ctypes declares argtypes -- and calls the reference's compiled code.  Nothing of the reference is
// compiled or copied.  The CUDA calls inside fail cleanly (no NVIDIA driver: "GPUassert: CUDA driver
// version is insufficient"), which is harmless because CSR keeps its arrays in host std::vectors and
// PCSR fills host ("pinned") arrays that this shim allocates itself before build_csr() runs.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <tuple>
#include <vector>

#include <fcntl.h>
#include <unistd.h>

// ---- csr.cu:35-59 ------------------------------------------------------------------------------
class CSR {
public:
    std::vector<int> row_offset, column_indices, eids, node_ids;
    int *dev[4];
    std::vector<int> in_degrees, out_degrees;
    std::vector<float> weighted_out_degrees;
    std::uintptr_t ptr[4];
    CSR(std::vector<std::tuple<int, int, int>> edge_list, std::vector<float> edge_weight, int num_nodes,
        bool is_edge_reverse);
};

// ---- pcsr.cu:44-89, 273-318 --------------------------------------------------------------------
struct node_t { uint32_t beginning, end, num_neighbors, in_degree; };
struct edge_t { uint32_t dest, value; };
struct edge_list_t { int N, H, logN; std::vector<edge_t> items; };
class PCSR {
public:
    std::vector<node_t> nodes;
    std::vector<uint32_t> in_degrees, out_degrees;
    edge_list_t edges;
    uint32_t edge_count;
    uint32_t *pinned[4];      // row_offset, column_indices, eids, node_ids
    uint32_t *device[4];
    PCSR(uint32_t init_n, uint32_t max_edge_count);
    void edge_update_list(std::vector<std::tuple<uint32_t, uint32_t>> edge_list, bool is_delete, bool is_reverse_edge);
    void label_edges();
    float build_csr();
    float build_reverse_csr();
    std::vector<std::tuple<uint32_t, uint32_t, uint32_t>> get_edges();
};

namespace {
struct Mute {                     // the reference printf()s one GPUassert line per failed CUDA call
    int saved;
    Mute() { fflush(stdout); saved = dup(1); int n = open("/dev/null", O_WRONLY); dup2(n, 1); close(n); }
    ~Mute() { fflush(stdout); dup2(saved, 1); close(saved); }
};
struct Handle { PCSR *p; size_t cap_e, n; };
}  // namespace

extern "C" {

int ref_csr_ctor(const int32_t *a, const int32_t *b, const int32_t *eid, const float *w, int64_t E, int32_t N,
                 int reverse, int32_t *row_offset, int32_t *col, int32_t *eids, int32_t *node_ids, int32_t *in_deg,
                 int32_t *out_deg, float *wdeg)
{
    Mute m;
    std::vector<std::tuple<int, int, int>> el((size_t)E);
    for (int64_t i = 0; i < E; ++i) el[i] = std::make_tuple(a[i], b[i], eid[i]);
    std::vector<float> ew(w, w + E);
    CSR c(el, ew, N, reverse != 0);
    if (c.row_offset.size() != (size_t)N + 1 || c.column_indices.size() != (size_t)E || c.eids.size() != (size_t)E ||
        c.node_ids.size() != (size_t)N || c.in_degrees.size() != (size_t)N || c.out_degrees.size() != (size_t)N ||
        c.weighted_out_degrees.size() != (size_t)N)
        return 1;
    memcpy(row_offset, c.row_offset.data(), sizeof(int) * (N + 1));
    memcpy(col, c.column_indices.data(), sizeof(int) * E);
    memcpy(eids, c.eids.data(), sizeof(int) * E);
    memcpy(node_ids, c.node_ids.data(), sizeof(int) * N);
    memcpy(in_deg, c.in_degrees.data(), sizeof(int) * N);
    memcpy(out_deg, c.out_degrees.data(), sizeof(int) * N);
    memcpy(wdeg, c.weighted_out_degrees.data(), sizeof(float) * N);
    return 0;
}

void *ref_pcsr_new(uint32_t n, uint32_t max_edges)
{
    Mute m;
    Handle *h = new Handle;
    h->p = new PCSR(n, max_edges);
    h->n = n;
    h->cap_e = (size_t)max_edges + 64;
    // cudaMallocHost failed inside the constructor: give build_csr() host arrays of the sizes it asked for
    h->p->pinned[0] = (uint32_t *)calloc(n + 1, 4);
    h->p->pinned[1] = (uint32_t *)calloc(h->cap_e, 4);
    h->p->pinned[2] = (uint32_t *)calloc(h->cap_e, 4);
    h->p->pinned[3] = (uint32_t *)calloc(n + 1, 4);
    for (int i = 0; i < 4; ++i) h->p->device[i] = nullptr;
    return h;
}

void *ref_pcsr_copy(void *hv)      // PCSR(self): what __deepcopy__ does (pcsr.cu:933-938); shares the output arrays
{
    Handle *h = (Handle *)hv, *c = new Handle(*h);
    c->p = new PCSR(*h->p);
    return c;
}

void ref_pcsr_update(void *hv, const uint32_t *a, const uint32_t *b, int64_t n, int is_delete, int is_reverse)
{
    Mute m;
    std::vector<std::tuple<uint32_t, uint32_t>> el((size_t)n);
    for (int64_t i = 0; i < n; ++i) el[i] = std::make_tuple(a[i], b[i]);
    ((Handle *)hv)->p->edge_update_list(el, is_delete != 0, is_reverse != 0);
}

void ref_pcsr_label(void *hv) { ((Handle *)hv)->p->label_edges(); }

int64_t ref_pcsr_edge_count(void *hv) { return ((Handle *)hv)->p->edge_count; }

// returns edge_count, or -1 if it exceeds what the reference allocated (it would have overflowed)
int64_t ref_pcsr_build(void *hv, int reverse, uint32_t *ro, uint32_t *col, uint32_t *eids, uint32_t *nid)
{
    Mute m;
    Handle *h = (Handle *)hv;
    if (h->p->edge_count > h->cap_e - 64) return -1;
    if (reverse) h->p->build_reverse_csr(); else h->p->build_csr();
    const size_t E = h->p->edge_count;
    memcpy(ro, h->p->pinned[0], 4 * (h->n + 1));
    memcpy(col, h->p->pinned[1], 4 * E);
    memcpy(eids, h->p->pinned[2], 4 * E);
    memcpy(nid, h->p->pinned[3], 4 * h->n);
    return (int64_t)E;
}

void ref_pcsr_degrees(void *hv, uint32_t *in_deg, uint32_t *out_deg)
{
    Handle *h = (Handle *)hv;
    memcpy(in_deg, h->p->in_degrees.data(), 4 * h->n);
    memcpy(out_deg, h->p->out_degrees.data(), 4 * h->n);
}

// PMA internals, to pin the restatement state for state: dims = {N, H, logN}; items as (dest, value) pairs
int64_t ref_pcsr_capacity(void *hv) { return ((Handle *)hv)->p->edges.N; }

void ref_pcsr_state(void *hv, int32_t *dims, uint32_t *items, uint32_t *nodes)
{
    PCSR *p = ((Handle *)hv)->p;
    dims[0] = p->edges.N; dims[1] = p->edges.H; dims[2] = p->edges.logN;
    memcpy(items, p->edges.items.data(), sizeof(edge_t) * p->edges.items.size());
    memcpy(nodes, p->nodes.data(), sizeof(node_t) * p->nodes.size());
}

int64_t ref_pcsr_get_edges(void *hv, uint32_t *out3)
{
    auto v = ((Handle *)hv)->p->get_edges();
    for (size_t i = 0; i < v.size(); ++i) {
        out3[3 * i] = std::get<0>(v[i]); out3[3 * i + 1] = std::get<1>(v[i]); out3[3 * i + 2] = std::get<2>(v[i]);
    }
    return (int64_t)v.size();
}

}  // extern "C"
