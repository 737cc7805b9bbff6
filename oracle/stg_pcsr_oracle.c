/* CPU oracle, part 2: the reference's PCSR dynamic-graph store.
 *
 * TEST INFRASTRUCTURE ONLY (same rule as stg_oracle.c): tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may use it; the product never does.
 *
 * Restates, function by function, reference stgraph/graph/dynamic/pcsr/pcsr.cu (a host-only packed
 * memory array: one sorted, gapped edge array with a sentinel per vertex).  Written from the
 * algorithm, in C, with explicit arrays instead of std::vector; every function cites the lines it
 * follows.  PINNED: tests/test_oracle_pcsr.py drives this file and the reference's own compiled
 * pcsr.so (through oracle/_ref/libref_shim.so) with the same random update streams and requires the
 * PMA state (items, nodes, N/H/logN), the degree counters and the emitted CSR arrays to be identical
 * after every step; tests/golden/pcsr_*.npz hold streams + outputs recorded from pcsr.so for the
 * places where /root/reference does not exist.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint32_t beginning, end, num_neighbors, in_degree; } pnode;   /* pcsr.cu:44-59 */
typedef struct { uint32_t dest, value; } pedge;                                /* pcsr.cu:61-72 */

typedef struct {
    pnode *nodes; size_t n_nodes, cap_nodes;
    uint32_t *in_degrees, *out_degrees; size_t n_deg;
    pedge *items; int N, H, logN;                                               /* edge_list_t, pcsr.cu:74-89 */
    uint32_t edge_count;
} pcsr;

/* index of the highest set bit (the reference's inline-asm `bsr`, pcsr.cu:126-133) */
static int bsr_word(int w) { int r = 0; unsigned u = (unsigned)w; while (u >>= 1) ++r; return r; }

static int is_null(pedge e) { return e.value == 0; }                                           /* pcsr.cu:140 */
static int is_sentinel(pedge e) { return e.dest == UINT32_MAX || e.value == UINT32_MAX; }      /* pcsr.cu:142-145 */
static int find_leaf(const pcsr *p, int index) { return (index / p->logN) * p->logN; }         /* pcsr.cu:136-139 */
static int find_node(int index, int len) { return (index / len) * len; }                       /* pcsr.cu:245 */

static void set_dims(pcsr *p)                                  /* pcsr.cu:326-328 and 466-467 */
{
    p->logN = 1 << bsr_word(bsr_word(p->N) + 1);
    p->H = bsr_word(p->N / p->logN);
}

/* pcsr.cu:148-229: position of the first edge with dest >= elem.dest inside (start, end), skipping gaps */
static uint32_t binary_search(const pcsr *p, const pedge *elem, uint32_t start, uint32_t end)
{
    while (start + 1 < end) {
        uint32_t mid = (start + end) / 2;
        pedge item = p->items[mid];
        uint32_t change = 1, check = mid;
        int flag = 1;
        while (is_null(item) && flag) {
            flag = 0;
            check = mid + change;
            if (check < end) {
                flag = 1;
                item = p->items[check];
                if (!is_null(item)) break;
            }
            check = mid - change;
            if (check >= start) {
                flag = 1;
                item = p->items[check];
            }
            change++;
        }
        if (is_null(item) || start == check || end == check) {
            if (!is_null(item) && start == check && elem->dest <= item.dest) return check;
            return mid;
        }
        if (elem->dest == item.dest) return check;
        if (elem->dest < item.dest) end = check; else start = check;
    }
    if (end < start) start = end;
    if (elem->dest <= p->items[start].dest && !is_null(p->items[start])) return start;
    return end;
}

static double get_density(const pcsr *p, int index, int len)                    /* pcsr.cu:232-243 */
{
    int full = 0;
    for (int i = index; i < index + len; i++) full += !is_null(p->items[i]);
    return (double)full / len;
}

static double density_upper(const pcsr *p, int depth) { return 3.0 / 4.0 + ((.25 * depth) / p->H); }   /* pcsr.cu:247-257 (.y) */

static uint32_t find_elem_pointer(const pcsr *p, uint32_t index, pedge elem)    /* pcsr.cu:264-272 */
{
    while (!(p->items[index].dest == elem.dest && p->items[index].value == elem.value)) ++index;
    return index;
}

static void fix_sentinel(pcsr *p, int32_t node_index, int in)                   /* pcsr.cu:603-614 */
{
    p->nodes[node_index].beginning = (uint32_t)in;
    if (node_index > 0) p->nodes[node_index - 1].end = (uint32_t)in;
    if ((size_t)node_index == p->n_nodes - 1) p->nodes[node_index].end = (uint32_t)(p->N - 1);
}

static void fix_if_sentinel(pcsr *p, pedge el, int index)       /* the block repeated at pcsr.cu:511-520 etc. */
{
    if (!is_null(el) && is_sentinel(el)) fix_sentinel(p, el.value == UINT32_MAX ? 0 : (int32_t)el.value, index);
}

static void redistribute(pcsr *p, int index, int len)                           /* pcsr.cu:616-655 */
{
    pedge *space = (pedge *)calloc((size_t)len, sizeof(pedge));
    int j = 0;
    for (int i = index; i < index + len; i++) {
        space[j] = p->items[i];
        j += !is_null(p->items[i]);
        p->items[i].value = 0;
        p->items[i].dest = 0;
    }
    double index_d = index, step = ((double)len) / j;
    for (int i = 0; i < j; i++) {
        int in = (int)index_d;
        p->items[in] = space[i];
        if (is_sentinel(space[i])) fix_sentinel(p, space[i].value == UINT32_MAX ? 0 : (int32_t)space[i].value, in);
        index_d += step;
    }
    free(space);
}

static void double_list(pcsr *p)                                                /* pcsr.cu:463-477 */
{
    p->N *= 2;
    set_dims(p);
    p->items = (pedge *)realloc(p->items, sizeof(pedge) * (size_t)p->N);
    memset(p->items + p->N / 2, 0, sizeof(pedge) * (size_t)(p->N / 2));
    redistribute(p, 0, p->N);
}

static void slide_left(pcsr *p, int index);

static int slide_right(pcsr *p, int index)                                      /* pcsr.cu:479-551 */
{
    int rval = 0;
    pedge el = p->items[index];
    p->items[index].dest = 0;
    p->items[index].value = 0;
    index++;
    while (index < p->N && !is_null(p->items[index])) {
        pedge temp = p->items[index];
        p->items[index] = el;
        fix_if_sentinel(p, el, index);
        el = temp;
        index++;
    }
    fix_if_sentinel(p, el, index);
    if (index == p->N) {
        index--;
        slide_left(p, index);
        rval = -1;
    }
    p->items[index] = el;
    return rval;
}

static void slide_left(pcsr *p, int index)                                      /* pcsr.cu:553-601 */
{
    pedge el = p->items[index];
    p->items[index].dest = 0;
    p->items[index].value = 0;
    index--;
    while (index >= 0 && !is_null(p->items[index])) {
        pedge temp = p->items[index];
        p->items[index] = el;
        fix_if_sentinel(p, el, index);
        el = temp;
        index--;
    }
    if (index == -1) {
        double_list(p);
        slide_right(p, 0);
        index = 0;
    }
    fix_if_sentinel(p, el, index);
    p->items[index] = el;
}

static uint32_t insert(pcsr *p, uint32_t index, pedge elem, uint32_t src)       /* pcsr.cu:386-461 */
{
    int node_index = find_leaf(p, (int)index);
    int level = p->H;
    int len = p->logN;

    if (is_null(p->items[index])) {
        p->items[index] = elem;
    } else {
        if (!is_sentinel(elem) && p->items[index].dest == elem.dest) {     /* existing edge: overwrite its value */
            p->items[index].value = elem.value;
            return index;
        }
        if (index == (uint32_t)(p->N - 1)) {
            double_list(p);
            pnode node = p->nodes[src];
            return insert(p, binary_search(p, &elem, node.beginning + 1, node.end), elem, src);
        }
        if (slide_right(p, (int)index) == -1) {
            index -= 1;
            slide_left(p, (int)index);
        }
        p->items[index] = elem;
    }

    double density = get_density(p, node_index, len);
    if (density == 1) {
        node_index = find_node(node_index, len * 2);
        redistribute(p, node_index, len * 2);
    } else {
        redistribute(p, node_index, len);
    }

    double upper = density_upper(p, level);
    density = get_density(p, node_index, len);
    while (density >= upper) {
        len *= 2;
        if (len <= p->N) {
            level--;
            node_index = find_node(node_index, len);
            upper = density_upper(p, level);
            density = get_density(p, node_index, len);
        } else {
            double_list(p);
            return find_elem_pointer(p, 0, elem);
        }
    }
    redistribute(p, node_index, len);
    return find_elem_pointer(p, (uint32_t)node_index, elem);
}

static void add_node(pcsr *p)                                                   /* pcsr.cu:358-384 */
{
    pnode node = {0, 0, 0, 0};
    size_t len = p->n_nodes;
    pedge sentinel = {UINT32_MAX, (uint32_t)len};
    if (len > 0) {
        node.beginning = p->nodes[len - 1].end;
        node.end = node.beginning + 1;
    } else {
        node.beginning = 0;
        node.end = 1;
        sentinel.value = UINT32_MAX;
    }
    if (p->n_nodes == p->cap_nodes) {
        p->cap_nodes = p->cap_nodes ? 2 * p->cap_nodes : 16;
        p->nodes = (pnode *)realloc(p->nodes, sizeof(pnode) * p->cap_nodes);
    }
    p->nodes[p->n_nodes++] = node;
    insert(p, node.beginning, sentinel, (uint32_t)(p->n_nodes - 1));
}

/* ---- exported ---------------------------------------------------------------------------------- */

void *orc_pcsr_new(uint32_t init_n, uint32_t max_num_edges)                     /* PCSR::PCSR, pcsr.cu:322-356 */
{
    (void)max_num_edges;                       /* only sizes the pinned/device output arrays in the reference */
    pcsr *p = (pcsr *)calloc(1, sizeof(pcsr));
    if (init_n != 0) {
        p->N = 2 << bsr_word((int)init_n);
        set_dims(p);
        p->items = (pedge *)calloc((size_t)p->N, sizeof(pedge));
        for (uint32_t i = 0; i < init_n; i++) add_node(p);
        p->in_degrees = (uint32_t *)calloc(init_n, 4);
        p->out_degrees = (uint32_t *)calloc(init_n, 4);
        p->n_deg = init_n;
    }
    return p;
}

void *orc_pcsr_copy(const void *pv)                                             /* PCSR(self), pcsr.cu:933-938 */
{
    const pcsr *p = (const pcsr *)pv;
    pcsr *c = (pcsr *)malloc(sizeof(pcsr));
    *c = *p;
    c->cap_nodes = p->n_nodes;
    c->nodes = (pnode *)malloc(sizeof(pnode) * (p->n_nodes ? p->n_nodes : 1));
    memcpy(c->nodes, p->nodes, sizeof(pnode) * p->n_nodes);
    c->in_degrees = (uint32_t *)malloc(4 * (p->n_deg ? p->n_deg : 1));
    c->out_degrees = (uint32_t *)malloc(4 * (p->n_deg ? p->n_deg : 1));
    memcpy(c->in_degrees, p->in_degrees, 4 * p->n_deg);
    memcpy(c->out_degrees, p->out_degrees, 4 * p->n_deg);
    c->items = (pedge *)malloc(sizeof(pedge) * (size_t)(p->N ? p->N : 1));
    memcpy(c->items, p->items, sizeof(pedge) * (size_t)p->N);
    return c;
}

void orc_pcsr_free(void *pv)
{
    pcsr *p = (pcsr *)pv;
    if (!p) return;
    free(p->nodes); free(p->in_degrees); free(p->out_degrees); free(p->items); free(p);
}

static void add_edge(pcsr *p, uint32_t src, uint32_t dest, uint32_t value)      /* pcsr.cu:657-675 */
{
    if (value != 0) {
        pnode node = p->nodes[src];
        p->nodes[src].num_neighbors++;
        p->nodes[dest].in_degree++;
        pedge e = {dest, value};
        insert(p, binary_search(p, &e, node.beginning + 1, node.end), e, src);
        ++p->edge_count;
    }
}

static void delete_edge(pcsr *p, uint32_t src, uint32_t dest)                   /* pcsr.cu:700-716 */
{
    pedge e = {dest, 0};
    uint32_t loc = binary_search(p, &e, p->nodes[src].beginning + 1, p->nodes[src].end);
    if (!is_null(p->items[loc]) && p->items[loc].dest == dest) {
        p->items[loc].value = 0;
        p->nodes[src].num_neighbors -= 1;
        p->nodes[dest].in_degree -= 1;
        --p->edge_count;
    }
}

/* pcsr.cu:759-779: the degree counters move even when the PMA itself ignores the update */
void orc_pcsr_edge_update_list(void *pv, const uint32_t *a, const uint32_t *b, int64_t n, int is_delete,
                               int is_reverse_edge)
{
    pcsr *p = (pcsr *)pv;
    for (int64_t i = 0; i < n; ++i) {
        uint32_t src = is_reverse_edge ? b[i] : a[i];
        uint32_t dst = is_reverse_edge ? a[i] : b[i];
        if (is_delete) {
            p->in_degrees[dst] -= 1;
            p->out_degrees[src] -= 1;
            delete_edge(p, src, dst);
        } else {
            p->in_degrees[dst] += 1;
            p->out_degrees[src] += 1;
            add_edge(p, src, dst, 1);
        }
    }
}

void orc_pcsr_label_edges(void *pv)                                             /* pcsr.cu:745-757: 1-based, array order */
{
    pcsr *p = (pcsr *)pv;
    uint32_t counter = 1;
    for (int i = 0; i < p->N; ++i)
        if (!is_sentinel(p->items[i]) && !is_null(p->items[i])) p->items[i].value = counter++;
}

typedef struct { uint32_t deg, id; } deg_id;
static int by_degree_desc(const void *x, const void *y)
{
    const deg_id *a = (const deg_id *)x, *b = (const deg_id *)y;
    if ((int)a->deg != (int)b->deg) return (int)a->deg > (int)b->deg ? -1 : 1;   /* compared as int (pcsr.cu:812-814) */
    return a->id < b->id ? -1 : (a->id > b->id);       /* ties: unspecified in the reference (std::sort); by id here */
}

static void node_ids_by_degree(const uint32_t *deg, size_t n, uint32_t *node_ids)   /* pcsr.cu:806-819 / 858-871 */
{
    deg_id *v = (deg_id *)malloc(sizeof(deg_id) * (n ? n : 1));
    for (size_t i = 0; i < n; ++i) { v[i].deg = deg[i]; v[i].id = (uint32_t)i; }
    qsort(v, n, sizeof(deg_id), by_degree_desc);
    for (size_t i = 0; i < n; ++i) node_ids[i] = v[i].id;
    free(v);
}

/* PCSR::build_csr, pcsr.cu:829-879: rows = PMA sources; a row is filled BACK TO FRONT, so its columns
 * come out in descending order; eids are the 1-based labels.  Returns edge_count. */
int64_t orc_pcsr_build_csr(void *pv, uint32_t *row_offset, uint32_t *column_indices, uint32_t *eids,
                           uint32_t *node_ids)
{
    pcsr *p = (pcsr *)pv;
    size_t n = p->n_nodes;
    row_offset[0] = p->out_degrees[0];
    for (size_t i = 1; i < p->n_deg; ++i) row_offset[i] = row_offset[i - 1] + p->out_degrees[i];
    row_offset[p->n_deg] = p->edge_count;
    for (size_t i = 0; i < n; i++)
        for (uint32_t j = p->nodes[i].beginning + 1; j < p->nodes[i].end; j++)
            if (!is_sentinel(p->items[j]) && !is_null(p->items[j])) {
                row_offset[i] -= 1;
                column_indices[row_offset[i]] = p->items[j].dest;
                eids[row_offset[i]] = p->items[j].value;
            }
    node_ids_by_degree(p->out_degrees, p->n_deg, node_ids);
    return p->edge_count;
}

/* PCSR::build_reverse_csr, pcsr.cu:781-827: rows = PMA destinations, again filled back to front */
int64_t orc_pcsr_build_reverse_csr(void *pv, uint32_t *row_offset, uint32_t *column_indices, uint32_t *eids,
                                   uint32_t *node_ids)
{
    pcsr *p = (pcsr *)pv;
    size_t n = p->n_nodes;
    row_offset[0] = p->in_degrees[0];
    for (size_t i = 1; i < p->n_deg; ++i) row_offset[i] = row_offset[i - 1] + p->in_degrees[i];
    row_offset[p->n_deg] = p->edge_count;
    for (size_t i = 0; i < n; i++)
        for (uint32_t j = p->nodes[i].beginning + 1; j < p->nodes[i].end; j++)
            if (!is_sentinel(p->items[j]) && !is_null(p->items[j])) {
                uint32_t d = p->items[j].dest;
                row_offset[d] -= 1;
                column_indices[row_offset[d]] = (uint32_t)i;
                eids[row_offset[d]] = p->items[j].value;
            }
    node_ids_by_degree(p->in_degrees, p->n_deg, node_ids);
    return p->edge_count;
}

int64_t orc_pcsr_edge_count(const void *pv) { return ((const pcsr *)pv)->edge_count; }
int64_t orc_pcsr_capacity(const void *pv) { return ((const pcsr *)pv)->N; }

void orc_pcsr_degrees(const void *pv, uint32_t *in_deg, uint32_t *out_deg)
{
    const pcsr *p = (const pcsr *)pv;
    memcpy(in_deg, p->in_degrees, 4 * p->n_deg);
    memcpy(out_deg, p->out_degrees, 4 * p->n_deg);
}

void orc_pcsr_state(const void *pv, int32_t *dims, uint32_t *items, uint32_t *nodes)
{
    const pcsr *p = (const pcsr *)pv;
    dims[0] = p->N; dims[1] = p->H; dims[2] = p->logN;
    memcpy(items, p->items, sizeof(pedge) * (size_t)p->N);
    memcpy(nodes, p->nodes, sizeof(pnode) * p->n_nodes);
}

int64_t orc_pcsr_get_edges(const void *pv, uint32_t *out3)                      /* pcsr.cu:723-743 */
{
    const pcsr *p = (const pcsr *)pv;
    int64_t k = 0;
    for (size_t i = 0; i < p->n_nodes; i++)
        for (uint32_t j = p->nodes[i].beginning + 1; j < p->nodes[i].end; j++)
            if (!is_null(p->items[j])) {
                out3[3 * k] = (uint32_t)i; out3[3 * k + 1] = p->items[j].dest; out3[3 * k + 2] = p->items[j].value;
                ++k;
            }
    return k;
}
