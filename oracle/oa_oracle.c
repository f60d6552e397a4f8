/*
 * oa_oracle.c -- CPU restatement of the reference ICP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under object_alignment_amd/ may link,
 * import or call this file.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, as the checker / the timed CPU baseline.
 *
 * Parity status: PINNED for everything that lives in /root/reference
 * (make_pairs' own logic and affine_matrix_from_points) against golden
 * fixtures produced by importing the reference (tools/gen_golden.py ->
 * tests/golden/ npz files).  UNPINNED for the part of the path that lives in
 * Blender's C code (mathutils float32 arithmetic, BVHTree.find_nearest):
 * Blender is not vendored in the reference and is not installed here, so the
 * float32 semantics below are a documented restatement ("Blender API
 * knowledge") and the correspondence rule is nearest *vertex* (SURVEY.md D2).
 *
 * Reference lines followed (all under /root/reference):
 *   functions/general.py:257-329   make_pairs
 *   functions/general.py:146-167,179-190,208-216   affine_matrix_from_points,
 *                                  live branch shear=False, usesvd=True
 *   operators/icp_align.py:82-151  the iterate loop and its 5-slot
 *                                  convergence ring
 *
 * Arithmetic conventions (shared bit-for-bit with the HIP path):
 *   M4 @ v3   : per row, acc(double) += (double)(float)(m*v), v.w = 1,
 *               result cast to float            (mathutils column_vector_multiplication)
 *   M4 @ M4   : same per element, k = 0..3     (mathutils matrix_mul)
 *   v.length  : acc(double) += (double)(float)(v[i]*v[i]) for i = 2,1,0; sqrt (double)
 *   inverted  : Blender's float adjoint / float determinant, element by element
 *               divided by the determinant (mathutils matrix_invert_internal,
 *               blenlib adjoint_m4_m4 / determinant_m4)
 *   NN metric : dx=qx-px (float) ...; d2 = fmaf(dz,dz, fmaf(dy,dy, dx*dx));
 *               argmin over target vertices, lowest index wins ties.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define OO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* float32 "mathutils" arithmetic                                      */
/* ------------------------------------------------------------------ */

/* functions/general.py:287,299,304 use Matrix @ Vector */
OO_API void oo_mat4_mul_vec3(const float *M, const float *v, float *out)
{
    float vc[4] = { v[0], v[1], v[2], 1.0f };
    float r[3];
    for (int row = 0; row < 3; ++row) {
        double acc = 0.0;
        for (int c = 0; c < 4; ++c) {
            float p = M[row * 4 + c] * vc[c];
            acc += (double)p;
        }
        r[row] = (float)acc;
    }
    out[0] = r[0]; out[1] = r[1]; out[2] = r[2];
}

/* operators/icp_align.py:121  matrix_world @ new_mat */
OO_API void oo_mat4_mul(const float *A, const float *B, float *out)
{
    float r[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 4; ++k) {
                float p = A[i * 4 + k] * B[k * 4 + j];
                acc += (double)p;
            }
            r[i * 4 + j] = (float)acc;
        }
    memcpy(out, r, sizeof r);
}

/* functions/general.py:265-266  mx.inverted().  Returns 1 on success, 0 if singular.
 *
 * Blender 3.2 (bl_info, /root/reference/__init__.py:22), source/blender/python/mathutils/mathutils_Matrix.c: Matrix_inverted ->
 * matrix_invert_internal: det = matrix_determinant_internal (4x4: determinant_m4), then matrix_invert_with_det_n_internal:
 * adjoint_matrix_n (4x4: adjoint_m4_m4) and every element divided by det -- ALL in C float (blenlib math_matrix.c:
 * determinant_m2 = a d - b c; determinant_m3 = a1 m2(b2,b3,c2,c3) - b1 m2(a2,a3,c2,c3) + c1 m2(a2,a3,b2,b3); determinant_m4 and
 * adjoint_m4_m4 expand along the first index with alternating signs), evaluated left to right, every operation rounded to float
 * (x86-64 build: no fused multiply-add; an arm64 build of Blender may contract a*b - c*d and differ in the last bit).  mathutils
 * stores matrices column-major (matrix[col * 4 + row]); the routines below work on that layout, as Blender's do, so the operand
 * ORDER inside every cofactor is Blender's.  Restated from the public source (Blender API knowledge); rounds 1-4
 * used an adjugate in double rounded once to float (kept as rule 1 for the test that measures the difference). */
static int g_inverse_rule = 0;       /* 0: Blender's float adjoint / float determinant; 1: rounds 1-4 (double, rounded once) */
OO_API void oo_set_inverse_rule(int rule) { g_inverse_rule = rule; }

static float bl_det_m2(float a, float b, float c, float d) { return a * d - b * c; }
static float bl_det_m3(float a1, float a2, float a3, float b1, float b2, float b3, float c1, float c2, float c3)
{
    float ans = (a1 * bl_det_m2(b2, b3, c2, c3) - b1 * bl_det_m2(a2, a3, c2, c3) + c1 * bl_det_m2(a2, a3, b2, b3));
    return ans;
}

static int mat4_inverted_double(const float *Af, float *out)
{
    double a[16];
    for (int i = 0; i < 16; ++i) a[i] = (double)Af[i];
    double s0 = a[0] * a[5] - a[4] * a[1];
    double s1 = a[0] * a[6] - a[4] * a[2];
    double s2 = a[0] * a[7] - a[4] * a[3];
    double s3 = a[1] * a[6] - a[5] * a[2];
    double s4 = a[1] * a[7] - a[5] * a[3];
    double s5 = a[2] * a[7] - a[6] * a[3];
    double c5 = a[10] * a[15] - a[14] * a[11];
    double c4 = a[9] * a[15] - a[13] * a[11];
    double c3 = a[9] * a[14] - a[13] * a[10];
    double c2 = a[8] * a[15] - a[12] * a[11];
    double c1 = a[8] * a[14] - a[12] * a[10];
    double c0 = a[8] * a[13] - a[12] * a[9];
    double det = ((((s0 * c5 - s1 * c4) + s2 * c3) + s3 * c2) - s4 * c1) + s5 * c0;
    if (det == 0.0) return 0;
    double b[16];
    b[0]  = (( a[5] * c5 - a[6] * c4) + a[7] * c3) / det;
    b[1]  = ((-a[1] * c5 + a[2] * c4) - a[3] * c3) / det;
    b[2]  = (( a[13] * s5 - a[14] * s4) + a[15] * s3) / det;
    b[3]  = ((-a[9] * s5 + a[10] * s4) - a[11] * s3) / det;
    b[4]  = ((-a[4] * c5 + a[6] * c2) - a[7] * c1) / det;
    b[5]  = (( a[0] * c5 - a[2] * c2) + a[3] * c1) / det;
    b[6]  = ((-a[12] * s5 + a[14] * s2) - a[15] * s1) / det;
    b[7]  = (( a[8] * s5 - a[10] * s2) + a[11] * s1) / det;
    b[8]  = (( a[4] * c4 - a[5] * c2) + a[7] * c0) / det;
    b[9]  = ((-a[0] * c4 + a[1] * c2) - a[3] * c0) / det;
    b[10] = (( a[12] * s4 - a[13] * s2) + a[15] * s0) / det;
    b[11] = ((-a[8] * s4 + a[9] * s2) - a[11] * s0) / det;
    b[12] = ((-a[4] * c3 + a[5] * c1) - a[6] * c0) / det;
    b[13] = (( a[0] * c3 - a[1] * c1) + a[2] * c0) / det;
    b[14] = ((-a[12] * s3 + a[13] * s1) - a[14] * s0) / det;
    b[15] = (( a[8] * s3 - a[9] * s1) + a[10] * s0) / det;
    for (int i = 0; i < 16; ++i) out[i] = (float)b[i];
    return 1;
}

OO_API int oo_mat4_inverted(const float *Af, float *out)
{
    if (g_inverse_rule == 1) return mat4_inverted_double(Af, out);
    /* m[i][j] = Blender's matrix[i][j] = element (row j, column i) of the row-major input */
    float m[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) m[i][j] = Af[j * 4 + i];
    const float a1 = m[0][0], b1 = m[0][1], c1 = m[0][2], d1 = m[0][3];
    const float a2 = m[1][0], b2 = m[1][1], c2 = m[1][2], d2 = m[1][3];
    const float a3 = m[2][0], b3 = m[2][1], c3 = m[2][2], d3 = m[2][3];
    const float a4 = m[3][0], b4 = m[3][1], c4 = m[3][2], d4 = m[3][3];
    /* determinant_m4 */
    const float det = (a1 * bl_det_m3(b2, b3, b4, c2, c3, c4, d2, d3, d4) - b1 * bl_det_m3(a2, a3, a4, c2, c3, c4, d2, d3, d4) +
                       c1 * bl_det_m3(a2, a3, a4, b2, b3, b4, d2, d3, d4) - d1 * bl_det_m3(a2, a3, a4, b2, b3, b4, c2, c3, c4));
    if (det == 0.0f) return 0;
    /* adjoint_m4_m4 */
    float R[4][4];
    R[0][0] = bl_det_m3(b2, b3, b4, c2, c3, c4, d2, d3, d4);
    R[1][0] = -bl_det_m3(a2, a3, a4, c2, c3, c4, d2, d3, d4);
    R[2][0] = bl_det_m3(a2, a3, a4, b2, b3, b4, d2, d3, d4);
    R[3][0] = -bl_det_m3(a2, a3, a4, b2, b3, b4, c2, c3, c4);
    R[0][1] = -bl_det_m3(b1, b3, b4, c1, c3, c4, d1, d3, d4);
    R[1][1] = bl_det_m3(a1, a3, a4, c1, c3, c4, d1, d3, d4);
    R[2][1] = -bl_det_m3(a1, a3, a4, b1, b3, b4, d1, d3, d4);
    R[3][1] = bl_det_m3(a1, a3, a4, b1, b3, b4, c1, c3, c4);
    R[0][2] = bl_det_m3(b1, b2, b4, c1, c2, c4, d1, d2, d4);
    R[1][2] = -bl_det_m3(a1, a2, a4, c1, c2, c4, d1, d2, d4);
    R[2][2] = bl_det_m3(a1, a2, a4, b1, b2, b4, d1, d2, d4);
    R[3][2] = -bl_det_m3(a1, a2, a4, b1, b2, b4, c1, c2, c4);
    R[0][3] = -bl_det_m3(b1, b2, b3, c1, c2, c3, d1, d2, d3);
    R[1][3] = bl_det_m3(a1, a2, a3, c1, c2, c3, d1, d2, d3);
    R[2][3] = -bl_det_m3(a1, a2, a3, b1, b2, b3, d1, d2, d3);
    R[3][3] = bl_det_m3(a1, a2, a3, b1, b2, b3, c1, c2, c3);
    /* matrix_invert_with_det_n_internal: element by element / det, same layout */
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out[j * 4 + i] = R[i][j] / det;
    return 1;
}

/* functions/general.py:299  (...).length */
OO_API double oo_vec3_length(const float *v)
{
    double acc = 0.0;
    for (int i = 2; i >= 0; --i) {
        float p = v[i] * v[i];
        acc += (double)p;
    }
    return sqrt(acc);
}

/* the NN metric: fp32, difference form, two explicit fmas */
static inline float oo_d2(float px, float py, float pz, float qx, float qy, float qz)
{
    float dx = qx - px, dy = qy - py, dz = qz - pz;
    float t = dx * dx;
    t = fmaf(dy, dy, t);
    t = fmaf(dz, dz, t);
    return t;
}

OO_API float oo_dist2(const float *p, const float *q)
{
    return oo_d2(p[0], p[1], p[2], q[0], q[1], q[2]);
}

/* ------------------------------------------------------------------ */
/* nearest vertex: the definitional brute force                        */
/* stands in for base_bvh.find_nearest at functions/general.py:297     */
/* ------------------------------------------------------------------ */
OO_API void oo_nn_brute(const float *q, int64_t nq, const float *tgt, int64_t nt,
                        int64_t *idx, float *d2out)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nq; ++i) {
        float px = q[3 * i], py = q[3 * i + 1], pz = q[3 * i + 2];
        float best = INFINITY;
        int64_t bi = -1;
        for (int64_t j = 0; j < nt; ++j) {
            float d = oo_d2(px, py, pz, tgt[3 * j], tgt[3 * j + 1], tgt[3 * j + 2]);
            if (d < best) { best = d; bi = j; }   /* strict: lowest index wins ties */
        }
        idx[i] = bi;
        if (d2out) d2out[i] = best;
    }
}

/* ------------------------------------------------------------------ */
/* nearest vertex: KD-tree with fp32-exact pruning (same answers as     */
/* oo_nn_brute, including ties).  This is the "tree" stand-in for the   */
/* reference's BVH and the CPU baseline that bench.py times.            */
/* ------------------------------------------------------------------ */
#define KD_LEAF 12

typedef struct {
    int32_t axis;      /* -1 => leaf */
    float   split;
    int64_t lo, hi;    /* point range [lo,hi) in permuted arrays */
    int64_t left, right;
} kd_node;

typedef struct {
    int64_t  n;
    float   *x, *y, *z;   /* permuted coordinates */
    int64_t *id;          /* original index */
    kd_node *nodes;
    int64_t  n_nodes, cap_nodes;
} kd_tree;

static inline float kd_coord(const float *t, int64_t i, int a) { return t[3 * i + a]; }

static void kd_select(const float *t, int64_t *perm, int64_t lo, int64_t hi, int64_t k, int a)
{
    /* quickselect on perm[lo,hi) so perm[k] is the k-th by coord a (ties by index) */
    while (hi - lo > 1) {
        int64_t mid = lo + (hi - lo) / 2;
        /* median of three pivot */
        int64_t c0 = perm[lo], c1 = perm[mid], c2 = perm[hi - 1];
        float v0 = kd_coord(t, c0, a), v1 = kd_coord(t, c1, a), v2 = kd_coord(t, c2, a);
        int64_t pidx;
        if ((v0 <= v1 && v1 <= v2) || (v2 <= v1 && v1 <= v0)) pidx = mid;
        else if ((v1 <= v0 && v0 <= v2) || (v2 <= v0 && v0 <= v1)) pidx = lo;
        else pidx = hi - 1;
        float pv = kd_coord(t, perm[pidx], a);
        int64_t pi = perm[pidx];
        /* three-way partition on (value, index) */
        int64_t i = lo, lt = lo, gt = hi;
        while (i < gt) {
            float v = kd_coord(t, perm[i], a);
            int less = (v < pv) || (v == pv && perm[i] < pi);
            int more = (v > pv) || (v == pv && perm[i] > pi);
            if (less) { int64_t tmp = perm[i]; perm[i] = perm[lt]; perm[lt] = tmp; ++i; ++lt; }
            else if (more) { --gt; int64_t tmp = perm[i]; perm[i] = perm[gt]; perm[gt] = tmp; }
            else ++i;
        }
        if (k < lt) hi = lt;
        else if (k >= gt) lo = gt;
        else return;
    }
}

static int64_t kd_new_node(kd_tree *T)
{
    if (T->n_nodes == T->cap_nodes) {
        T->cap_nodes = T->cap_nodes ? T->cap_nodes * 2 : 1024;
        T->nodes = (kd_node *)realloc(T->nodes, (size_t)T->cap_nodes * sizeof(kd_node));
    }
    return T->n_nodes++;
}

static int64_t kd_build_rec(kd_tree *T, const float *t, int64_t *perm, int64_t lo, int64_t hi)
{
    int64_t me = kd_new_node(T);
    kd_node nd; nd.lo = lo; nd.hi = hi; nd.left = nd.right = -1; nd.axis = -1; nd.split = 0.f;
    if (hi - lo > KD_LEAF) {
        float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (int64_t i = lo; i < hi; ++i)
            for (int a = 0; a < 3; ++a) {
                float v = kd_coord(t, perm[i], a);
                if (v < mn[a]) mn[a] = v;
                if (v > mx[a]) mx[a] = v;
            }
        int a = 0;
        if (mx[1] - mn[1] > mx[a] - mn[a]) a = 1;
        if (mx[2] - mn[2] > mx[a] - mn[a]) a = 2;
        if (mx[a] > mn[a]) {
            int64_t mid = lo + (hi - lo) / 2;
            kd_select(t, perm, lo, hi, mid, a);
            nd.axis = a;
            nd.split = kd_coord(t, perm[mid], a);
            T->nodes[me] = nd;
            int64_t l = kd_build_rec(T, t, perm, lo, mid);
            int64_t r = kd_build_rec(T, t, perm, mid, hi);
            T->nodes[me].left = l;
            T->nodes[me].right = r;
            return me;
        }
    }
    T->nodes[me] = nd;
    return me;
}

OO_API void *oo_kd_build(const float *tgt, int64_t nt)
{
    kd_tree *T = (kd_tree *)calloc(1, sizeof(kd_tree));
    T->n = nt;
    int64_t *perm = (int64_t *)malloc((size_t)(nt > 0 ? nt : 1) * sizeof(int64_t));
    for (int64_t i = 0; i < nt; ++i) perm[i] = i;
    if (nt > 0) kd_build_rec(T, tgt, perm, 0, nt);
    T->x = (float *)malloc((size_t)(nt > 0 ? nt : 1) * sizeof(float));
    T->y = (float *)malloc((size_t)(nt > 0 ? nt : 1) * sizeof(float));
    T->z = (float *)malloc((size_t)(nt > 0 ? nt : 1) * sizeof(float));
    T->id = perm;
    for (int64_t i = 0; i < nt; ++i) {
        T->x[i] = tgt[3 * perm[i]];
        T->y[i] = tgt[3 * perm[i] + 1];
        T->z[i] = tgt[3 * perm[i] + 2];
    }
    return T;
}

OO_API void oo_kd_free(void *h)
{
    kd_tree *T = (kd_tree *)h;
    if (!T) return;
    free(T->x); free(T->y); free(T->z); free(T->id); free(T->nodes); free(T);
}

static void kd_query_one(const kd_tree *T, float px, float py, float pz, int64_t *bi_out, float *bd_out)
{
    float best = INFINITY;
    int64_t bi = -1;
    if (T->n == 0) { *bi_out = -1; *bd_out = best; return; }
    /* explicit stack of (node, lower bound) */
    int64_t stk[128]; float bnd[128]; int sp = 0;
    stk[sp] = 0; bnd[sp] = 0.f; ++sp;
    const float pc[3] = { px, py, pz };
    while (sp > 0) {
        --sp;
        int64_t ni = stk[sp];
        /* a subtree whose fp32 lower bound exceeds best cannot hold a better
           or an equal-distance point; equal bound must still be visited
           because a lower index could tie */
        if (bnd[sp] > best) continue;
        const kd_node *nd = &T->nodes[ni];
        while (nd->axis >= 0) {
            float c = pc[nd->axis];
            float diff = nd->split - c;           /* fl(s - c) */
            float b = diff * diff;                /* fl(fl(s-c)^2) <= computed d2 of any far-side point */
            int64_t nearc, farc;
            if (c < nd->split) { nearc = nd->left; farc = nd->right; }
            else if (c > nd->split) { nearc = nd->right; farc = nd->left; }
            else { nearc = nd->left; farc = nd->right; b = 0.f; }
            stk[sp] = farc; bnd[sp] = b; ++sp;
            nd = &T->nodes[nearc];
        }
        for (int64_t i = nd->lo; i < nd->hi; ++i) {
            float d = oo_d2(px, py, pz, T->x[i], T->y[i], T->z[i]);
            int64_t id = T->id[i];
            if (d < best || (d == best && id < bi)) { best = d; bi = id; }
        }
    }
    *bi_out = bi; *bd_out = best;
}

OO_API void oo_kd_query(const void *h, const float *q, int64_t nq, int64_t *idx, float *d2out, int nthreads)
{
    const kd_tree *T = (const kd_tree *)h;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1024) num_threads(nthreads)
#endif
    for (int64_t i = 0; i < nq; ++i) {
        int64_t bi; float bd;
        kd_query_one(T, q[3 * i], q[3 * i + 1], q[3 * i + 2], &bi, &bd);
        idx[i] = bi;
        if (d2out) d2out[i] = bd;
    }
}

OO_API int oo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ */
/* closest point on the triangle surface -- what BVHTree.find_nearest  */
/* (functions/general.py:297) actually returns in Blender.  The tree    */
/* itself is Blender C code outside /root/reference; this restates its  */
/* per-triangle callback (closest_on_tri_to_point_v3: Ericson,          */
/* "Real-Time Collision Detection" 5.1.5) in float32 with explicit      */
/* operation order and no fma, squared distance = |p - r|^2 in float32, */
/* nearest triangle wins, lowest triangle index on ties.                */
/* Blender API knowledge -- PARITY UNPINNED (no Blender here).          */
/* ------------------------------------------------------------------ */
static inline float dot3f(const float *a, const float *b)
{
    float s = a[0] * b[0];
    s = s + a[1] * b[1];
    s = s + a[2] * b[2];
    return s;
}

OO_API void oo_closest_on_tri(const float *p, const float *a, const float *b, const float *c, float *r)
{
    float ab[3], ac[3], ap[3], bp[3], cp[3];
    for (int i = 0; i < 3; ++i) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; ap[i] = p[i] - a[i]; }
    const float d1 = dot3f(ab, ap), d2 = dot3f(ac, ap);
    if (d1 <= 0.0f && d2 <= 0.0f) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; return; }          /* vertex A */
    for (int i = 0; i < 3; ++i) bp[i] = p[i] - b[i];
    const float d3 = dot3f(ab, bp), d4 = dot3f(ac, bp);
    if (d3 >= 0.0f && d4 <= d3) { r[0] = b[0]; r[1] = b[1]; r[2] = b[2]; return; }            /* vertex B */
    const float vc = d1 * d4 - d3 * d2;
    if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) {                                             /* edge AB */
        const float v = d1 / (d1 - d3);
        for (int i = 0; i < 3; ++i) r[i] = a[i] + ab[i] * v;
        return;
    }
    for (int i = 0; i < 3; ++i) cp[i] = p[i] - c[i];
    const float d5 = dot3f(ab, cp), d6 = dot3f(ac, cp);
    if (d6 >= 0.0f && d5 <= d6) { r[0] = c[0]; r[1] = c[1]; r[2] = c[2]; return; }            /* vertex C */
    const float vb = d5 * d2 - d1 * d6;
    if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) {                                             /* edge AC */
        const float w = d2 / (d2 - d6);
        for (int i = 0; i < 3; ++i) r[i] = a[i] + ac[i] * w;
        return;
    }
    const float va = d3 * d6 - d5 * d4;
    if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) {                               /* edge BC */
        const float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        for (int i = 0; i < 3; ++i) { float t = c[i] - b[i]; t = t * w; r[i] = t + b[i]; }
        return;
    }
    const float denom = 1.0f / ((va + vb) + vc);                                              /* face interior */
    const float v = vb * denom, w = vc * denom;
    for (int i = 0; i < 3; ++i) { const float acw = ac[i] * w; float t = a[i] + ab[i] * v; r[i] = t + acw; }
}

static inline float tri_dist2(const float *p, const float *r)
{
    float d[3] = { r[0] - p[0], r[1] - p[1], r[2] - p[2] };
    return dot3f(d, d);
}

/* definitional brute force over all triangles; face = -1 and d2 = +inf when there is no usable triangle */
OO_API void oo_nn_tri_brute(const float *q, int64_t nq, const float *verts, const int32_t *tris, int64_t ntris,
                            int64_t *face, float *co1, float *d2out)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nq; ++i) {
        float best = INFINITY, br[3] = { 0, 0, 0 };
        int64_t bf = -1;
        for (int64_t t = 0; t < ntris; ++t) {
            float r[3];
            oo_closest_on_tri(q + 3 * i, verts + 3 * (int64_t)tris[3 * t], verts + 3 * (int64_t)tris[3 * t + 1],
                              verts + 3 * (int64_t)tris[3 * t + 2], r);
            const float d = tri_dist2(q + 3 * i, r);
            if (d < best) { best = d; bf = t; br[0] = r[0]; br[1] = r[1]; br[2] = r[2]; }
        }
        if (face) face[i] = bf;
        if (co1) { co1[3 * i] = br[0]; co1[3 * i + 1] = br[1]; co1[3 * i + 2] = br[2]; }
        if (d2out) d2out[i] = best;
    }
}

/* ------------------------------------------------------------------ */
/* make_pairs  (functions/general.py:257-329)                          */
/* ------------------------------------------------------------------ */
/*
 * src      : n_verts x 3 float32, align-LOCAL coordinates (vert.co, :284)
 * vlist    : vertex indices to use (NULL => 0..n_verts-1), :259
 * sample   : stride, applied only when > 1 (:274-275)
 * kd       : tree over tgt (NULL => brute force); tgt is base-LOCAL
 * tris     : NULL => nearest target VERTEX; else n_tris x 3 vertex indices and the closest point on the
 *            triangle SURFACE is used (the BVHTree.find_nearest semantics)
 * mx1, mx2 : align / base matrix_world, row-major float32 (:262-263)
 * A, B     : caller-allocated 3 x cap row-major doubles (row = axis, :313-321)
 * returns K >= 0, or -1 if thresh <= 0 (reference falls through and returns
 *         None, :277), -2 singular matrix, -3 cap too small
 */
/* EXTENSION (not in the reference): angle between world-space normals, carried with the
 * inverse-transpose of the object matrices; same arithmetic as normal_angle_ok in oa_kernels.hpp */
static int normal_angle_ok(const float *imx1, const float *imx2, const float *ns, const float *nt, double cos_min)
{
    double a[3], b[3];
    for (int k = 0; k < 3; ++k) {
        a[k] = (double)imx1[k] * (double)ns[0] + (double)imx1[4 + k] * (double)ns[1] + (double)imx1[8 + k] * (double)ns[2];
        b[k] = (double)imx2[k] * (double)nt[0] + (double)imx2[4 + k] * (double)nt[1] + (double)imx2[8 + k] * (double)nt[2];
    }
    const double ab = (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
    const double aa = (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2];
    const double bb = (b[0] * b[0] + b[1] * b[1]) + b[2] * b[2];
    const double c = ab / sqrt(aa * bb);
    return c >= cos_min;
}

static void pair_normal(const float *tgt, const int32_t *tris, const float *tgt_n, int64_t idx, float *tn)
{
    if (tris) {                                  /* geometric face normal, Blender normal_tri_v3 operand order */
        const float *a = tgt + 3 * (int64_t)tris[3 * idx], *b = tgt + 3 * (int64_t)tris[3 * idx + 1], *c = tgt + 3 * (int64_t)tris[3 * idx + 2];
        const float e1[3] = { a[0] - b[0], a[1] - b[1], a[2] - b[2] };
        const float e2[3] = { b[0] - c[0], b[1] - c[1], b[2] - c[2] };
        tn[0] = e1[1] * e2[2] - e1[2] * e2[1];
        tn[1] = e1[2] * e2[0] - e1[0] * e2[2];
        tn[2] = e1[0] * e2[1] - e1[1] * e2[0];
    } else { tn[0] = tgt_n[3 * idx]; tn[1] = tgt_n[3 * idx + 1]; tn[2] = tgt_n[3 * idx + 2]; }
}

OO_API int64_t oo_make_pairs(const float *src, int64_t n_verts,
                             const int64_t *vlist, int64_t n_vlist, int sample,
                             const float *tgt, int64_t nt, const void *kd,
                             const int32_t *tris, int64_t n_tris,
                             const float *src_n, const float *tgt_n, double cos_min,
                             const float *mx1, const float *mx2,
                             double thresh, int calc_stats, int nthreads,
                             double *A, double *B, int64_t cap,
                             double *dstats, int64_t *nn_idx_out)
{
    if (!(thresh > 0)) return -1;
    float imx1[16], imx2[16];
    if (!oo_mat4_inverted(mx1, imx1)) return -2;
    if (!oo_mat4_inverted(mx2, imx2)) return -2;
    int64_t n_all = vlist ? n_vlist : n_verts;
    int64_t step = sample > 1 ? sample : 1;
    int64_t n_sel = (n_all + step - 1) / step;
    if (n_sel == 0) { if (calc_stats && dstats) { dstats[0] = NAN; dstats[1] = NAN; } return 0; }

    float *cof = (float *)malloc((size_t)n_sel * 3 * sizeof(float));
    int64_t *nn = (int64_t *)malloc((size_t)n_sel * sizeof(int64_t));
    /* co_find = imx2 @ (mx1 @ vert.co)   :287 */
#pragma omp parallel for schedule(static)
    for (int64_t s = 0; s < n_sel; ++s) {
        int64_t vi = vlist ? vlist[s * step] : s * step;
        float w[3];
        oo_mat4_mul_vec3(mx1, src + 3 * vi, w);
        oo_mat4_mul_vec3(imx2, w, cof + 3 * s);
    }
    float *co1buf = (float *)malloc((size_t)n_sel * 3 * sizeof(float));
    if (tris) {
        oo_nn_tri_brute(cof, n_sel, tgt, tris, n_tris, nn, co1buf, NULL);
    } else {
        if (kd) oo_kd_query(kd, cof, n_sel, nn, NULL, nthreads);
        else oo_nn_brute(cof, n_sel, tgt, nt, nn, NULL);
        for (int64_t s = 0; s < n_sel; ++s)
            for (int a = 0; a < 3; ++a) co1buf[3 * s + a] = nn[s] >= 0 ? tgt[3 * nn[s] + a] : 0.f;
    }

    int64_t K = 0;
    double sd = 0.0, sd2 = 0.0;
    for (int64_t s = 0; s < n_sel; ++s) {
        int64_t vi = vlist ? vlist[s * step] : s * step;
        if (nn_idx_out) nn_idx_out[s] = nn[s];
        if (nn[s] < 0) continue;
        const float *co1 = co1buf + 3 * s;
        float wa[3], wb[3], df[3];
        oo_mat4_mul_vec3(mx2, cof + 3 * s, wa);       /* mx2 @ co_find  :299 */
        oo_mat4_mul_vec3(mx2, co1, wb);               /* mx2 @ co1      :299 */
        df[0] = wa[0] - wb[0]; df[1] = wa[1] - wb[1]; df[2] = wa[2] - wb[2];
        double dist = oo_vec3_length(df);
        int keep = dist < thresh;                      /* :302 */
        if (keep && src_n) { float tn[3]; pair_normal(tgt, tris, tgt_n, nn[s], tn); keep = normal_angle_ok(imx1, imx2, src_n + 3 * vi, tn, cos_min); }
        if (keep) {
            if (K >= cap) { free(cof); free(nn); free(co1buf); return -3; }
            float b[3];
            oo_mat4_mul_vec3(imx1, wb, b);             /* imx1 @ (mx2 @ co1)  :304 */
            for (int a = 0; a < 3; ++a) {
                A[a * cap + K] = (double)src[3 * vi + a];   /* vert.co  :303 */
                B[a * cap + K] = (double)b[a];
            }
            sd += dist; sd2 += dist * dist;
            ++K;
        }
    }
    if (calc_stats && dstats) {
        if (K > 0) {
            /* two-pass population std, as np.std (:324-325) */
            double mean = sd / (double)K;
            dstats[0] = mean;
            double acc = 0.0;
            /* recompute distances for the second pass from A,B is not possible
               (they are align-local); redo the world distances */
            int64_t k = 0;
            for (int64_t s = 0; s < n_sel && k < K; ++s) {
                if (nn[s] < 0) continue;
                const float *co1 = co1buf + 3 * s;
                float wa[3], wb[3], df[3];
                oo_mat4_mul_vec3(mx2, cof + 3 * s, wa);
                oo_mat4_mul_vec3(mx2, co1, wb);
                df[0] = wa[0] - wb[0]; df[1] = wa[1] - wb[1]; df[2] = wa[2] - wb[2];
                double dist = oo_vec3_length(df);
                int keep = dist < thresh;
                if (keep && src_n) {
                    int64_t vi2 = vlist ? vlist[s * step] : s * step;
                    float tn[3]; pair_normal(tgt, tris, tgt_n, nn[s], tn);
                    keep = normal_angle_ok(imx1, imx2, src_n + 3 * vi2, tn, cos_min);
                }
                if (keep) { acc += (dist - mean) * (dist - mean); ++k; }
            }
            dstats[1] = sqrt(acc / (double)K);
        } else { dstats[0] = NAN; dstats[1] = NAN; }
    }
    (void)sd2;
    free(cof); free(nn); free(co1buf);
    return K;
}

/* ------------------------------------------------------------------ */
/* 3x3 SVD-based rotation (Kabsch), functions/general.py:179-190        */
/* ------------------------------------------------------------------ */
/* One-sided Jacobi on H (3x3): G = H V with orthogonal columns.
 * R = U diag(1,1,det) V^T equals [u1 u2 u1xu2][v1 v2 v1xv2]^T, which needs
 * only the two leading singular triplets (well defined for rank >= 2).   */
static void rot_from_H(const double H[9], double R[9])
{
    double G[9], V[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    memcpy(G, H, sizeof G);
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double app = 0, aqq = 0, apq = 0;
                for (int r = 0; r < 3; ++r) {
                    app += G[r * 3 + p] * G[r * 3 + p];
                    aqq += G[r * 3 + q] * G[r * 3 + q];
                    apq += G[r * 3 + p] * G[r * 3 + q];
                }
                if (apq == 0.0 || fabs(apq) <= 1e-17 * sqrt(app * aqq)) continue;
                rotated = 1;
                double zeta = (aqq - app) / (2.0 * apq);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int r = 0; r < 3; ++r) {
                    double gp = G[r * 3 + p], gq = G[r * 3 + q];
                    G[r * 3 + p] = c * gp - s * gq;
                    G[r * 3 + q] = s * gp + c * gq;
                    double vp = V[r * 3 + p], vq = V[r * 3 + q];
                    V[r * 3 + p] = c * vp - s * vq;
                    V[r * 3 + q] = s * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    double sg[3];
    int ord[3] = { 0, 1, 2 };
    for (int j = 0; j < 3; ++j)
        sg[j] = sqrt(G[j] * G[j] + G[3 + j] * G[3 + j] + G[6 + j] * G[6 + j]);
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (sg[ord[j]] > sg[ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    double u1[3], u2[3], u3[3], v1[3], v2[3], v3[3];
    int j1 = ord[0], j2 = ord[1];
    for (int r = 0; r < 3; ++r) {
        v1[r] = V[r * 3 + j1]; v2[r] = V[r * 3 + j2];
        u1[r] = sg[j1] > 0 ? G[r * 3 + j1] / sg[j1] : (r == 0);
        u2[r] = sg[j2] > 0 ? G[r * 3 + j2] / sg[j2] : 0.0;
    }
    if (!(sg[j2] > 0)) {
        /* rank <= 1: rotation not unique; pick any unit vector orthogonal to u1 / v1 pairing */
        int m = 0;
        if (fabs(u1[1]) < fabs(u1[m])) m = 1;
        if (fabs(u1[2]) < fabs(u1[m])) m = 2;
        double e[3] = { 0, 0, 0 }; e[m] = 1.0;
        double d = e[0] * u1[0] + e[1] * u1[1] + e[2] * u1[2];
        double n = 0;
        for (int r = 0; r < 3; ++r) { u2[r] = e[r] - d * u1[r]; n += u2[r] * u2[r]; }
        n = sqrt(n);
        for (int r = 0; r < 3; ++r) u2[r] /= n;
    }
    u3[0] = u1[1] * u2[2] - u1[2] * u2[1];
    u3[1] = u1[2] * u2[0] - u1[0] * u2[2];
    u3[2] = u1[0] * u2[1] - u1[1] * u2[0];
    v3[0] = v1[1] * v2[2] - v1[2] * v2[1];
    v3[1] = v1[2] * v2[0] - v1[0] * v2[2];
    v3[2] = v1[0] * v2[1] - v1[1] * v2[0];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            R[i * 3 + j] = u1[i] * v1[j] + u2[i] * v2[j] + u3[i] * v3[j];
}

/*
 * affine_matrix_from_points(v0=A, v1=B, shear=False, scale=with_scale, usesvd=True)
 * A, B : 3 x K row-major with leading dimension ld.  M : 4x4 row-major.
 * returns 0, or -1 for the reference's ValueError (K < 3; :150-157)
 */
OO_API int oo_kabsch(const double *A, const double *B, int64_t K, int64_t ld,
                     int with_scale, double *M)
{
    if (K < 3) return -1;
    double c0[3] = { 0, 0, 0 }, c1[3] = { 0, 0, 0 };
    for (int a = 0; a < 3; ++a) {
        double s0 = 0, s1 = 0;
        for (int64_t k = 0; k < K; ++k) { s0 += A[a * ld + k]; s1 += B[a * ld + k]; }
        c0[a] = s0 / (double)K;          /* -t0, :160 */
        c1[a] = s1 / (double)K;          /* -t1, :164 */
    }
    double H[9] = { 0 }, n0 = 0, n1 = 0;
    for (int64_t k = 0; k < K; ++k) {
        double a[3], b[3];
        for (int r = 0; r < 3; ++r) { a[r] = A[r * ld + k] - c0[r]; b[r] = B[r * ld + k] - c1[r]; }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) H[i * 3 + j] += b[i] * a[j];   /* dot(v1, v0.T), :181 */
        n0 += a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
        n1 += b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
    }
    double R[9];
    rot_from_H(H, R);                    /* :181-187 */
    double sc = 1.0;
    if (with_scale) sc = sqrt(n1 / n0);  /* :208-212 */
    for (int i = 0; i < 3; ++i) {
        double t = c1[i];
        for (int j = 0; j < 3; ++j) {
            M[i * 4 + j] = sc * R[i * 3 + j];
            t -= sc * R[i * 3 + j] * c0[j];
        }
        M[i * 4 + 3] = t;                /* inv(M1) . M . M0, :215 */
    }
    M[12] = M[13] = M[14] = 0.0; M[15] = 1.0;
    return 0;
}

/* ------------------------------------------------------------------ */
/* the operator's iterate loop  (operators/icp_align.py:91-151)         */
/* ------------------------------------------------------------------ */
typedef struct {
    int32_t iters;          /* icp_iterations                       */
    int32_t sample;         /* factor = round(1/sample_fraction)    */
    int32_t use_target;     /* calc_stats + convergence test        */
    int32_t with_scale;     /* align_meth == '1'                    */
    double  thresh;         /* min_start                            */
    double  target_d;       /* target_d                             */
} oo_settings;

typedef struct {
    int32_t iters_done;
    int32_t converged;
    int32_t status;         /* 0 ok, -1 thresh<=0, -2 singular, -4 fewer than 3 pairs */
    int32_t pad;
    double  last_translation;
    double  mean_dist, std_dist;
    int64_t last_K;
} oo_report;

/*
 * mx1 is updated in place (align_obj.matrix_world, :121).
 * step_M (iters x 16 doubles), step_new (iters x 16 floats), step_K, step_stats
 * (iters x 2) may be NULL.
 */
OO_API int oo_icp_run(const float *src, int64_t n_verts,
                      const int64_t *vlist, int64_t n_vlist,
                      const float *tgt, int64_t nt, const void *kd,
                      const int32_t *tris, int64_t n_tris,
                      const float *src_n, const float *tgt_n, double cos_min,
                      float *mx1, const float *mx2, const oo_settings *st, int nthreads,
                      oo_report *rep, double *step_M, float *step_new,
                      int64_t *step_K, double *step_stats, double *step_trans)
{
    int64_t n_all = vlist ? n_vlist : n_verts;
    int64_t step = st->sample > 1 ? st->sample : 1;
    int64_t cap = (n_all + step - 1) / step;
    if (cap < 1) cap = 1;
    double *A = (double *)malloc((size_t)cap * 3 * sizeof(double));
    double *B = (double *)malloc((size_t)cap * 3 * sizeof(double));
    double ring[5];
    for (int i = 0; i < 5; ++i) ring[i] = st->target_d * 2;      /* :93 */
    int n = 0, converged = 0;
    memset(rep, 0, sizeof *rep);
    while (n < st->iters && !converged) {                        /* :96 */
        double ds[2] = { NAN, NAN };
        int64_t K = oo_make_pairs(src, n_verts, vlist, n_vlist, st->sample, tgt, nt, kd, tris, n_tris, src_n, tgt_n, cos_min,
                                  mx1, mx2, st->thresh, st->use_target, nthreads,
                                  A, B, cap, ds, NULL);           /* :101 */
        if (K < 0) { rep->status = (int32_t)K; break; }
        double M[16];
        if (oo_kabsch(A, B, K, cap, st->with_scale, M) != 0) { rep->status = -4; break; }  /* :106-109 */
        float new_mat[16];
        for (int i = 0; i < 16; ++i) new_mat[i] = (float)M[i];   /* :116-119 */
        oo_mat4_mul(mx1, new_mat, mx1);                          /* :121 */
        float tr[3] = { new_mat[3], new_mat[7], new_mat[11] };   /* :129 */
        double tl = oo_vec3_length(tr);
        if (step_M) memcpy(step_M + 16 * n, M, sizeof M);
        if (step_new) memcpy(step_new + 16 * n, new_mat, sizeof new_mat);
        if (step_K) step_K[n] = K;
        if (step_stats) { step_stats[2 * n] = ds[0]; step_stats[2 * n + 1] = ds[1]; }
        if (step_trans) step_trans[n] = tl;
        rep->last_K = K; rep->mean_dist = ds[0]; rep->std_dist = ds[1];
        rep->last_translation = tl;
        if (st->use_target) {                                    /* :136 */
            ring[n % 5] = tl;                                    /* :137-138 */
            int all = 1;
            for (int i = 0; i < 5; ++i) if (!(ring[i] < st->target_d)) all = 0;
            if (all) converged = 1;                              /* :141-142 */
        }
        ++n;                                                     /* :151 */
    }
    rep->iters_done = n;
    rep->converged = converged;
    free(A); free(B);
    return rep->status;
}
