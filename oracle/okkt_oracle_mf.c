/*
 * okkt_oracle_mf.c -- CPU BASELINE / second oracle (test infrastructure, NOT product code).
 *
 * Only tests/ and bench.py's cpu_baseline leg may build, load or call this file.
 *
 * What it is.  The reference factors its KKT matrices with SuiteSparse CHOLMOD through Julia's stdlib
 * (/root/reference/src/linear_system_solvers/julia.jl:34,52): `cholesky` runs CHOLMOD's SUPERNODAL numeric phase (dense
 * BLAS-3 kernels on supernodes, one thread in the reference's published runs, docs/one-phase.tex:930), `ldlt` the
 * simplicial one.  CHOLMOD is not in /root/reference and cannot be built here, so the published algorithm class is
 * restated in plain C: a supernodal MULTIFRONTAL LDL^T without pivoting (Duff & Reid 1983; Liu, "The multifrontal
 * method for sparse matrix solution", SIAM Review 34, 1992) -- elimination tree, column counts, fundamental supernodes,
 * frontal assembly by extend-add, blocked dense partial factorisation, and OpenMP tasks over the elimination tree plus
 * inside the large fronts ("all host cores" of BASELINE.md section 2).  okkt_oracle.c is the simplicial counterpart.
 *
 * The permutation is an input (perm[new] = old), as for okkt_oracle.c: the parity tests hand it the product's ordering.
 * Inertia rule: julia.jl:70-90 (pos = d > tol, neg = d < -tol, zero otherwise, non-finite counted apart).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct mf_factor {
  int64_t n, nnz_in;
  int64_t *perm, *iperm;
  int64_t *parent, *count;      /* column elimination tree, column counts incl. diagonal */
  int64_t ns;                   /* supernodes */
  int64_t *sn0;                 /* [ns + 1] first column */
  int64_t *sparent;             /* [ns] */
  int64_t *child_ptr, *child;   /* children lists */
  int64_t *rp, *rows;           /* [ns + 1], front row lists (sorted, first k = own columns) */
  int64_t *lpos;                /* [ns + 1] offsets of the f x k panels in L */
  int64_t *amap_sn, *amap_off;  /* per input entry: supernode and offset (row_local + col_local * f), -1 = ignored */
  double *L, *D;
  double flops;
  int64_t max_front;
  /* numeric work */
  double **cb;                  /* [ns] contribution blocks (r x r, column-major, lower), alive between a front and its parent */
  int *pending;
} mf_factor;

void mf_free(mf_factor *F) {
  if (!F) return;
  free(F->perm); free(F->iperm); free(F->parent); free(F->count); free(F->sn0); free(F->sparent); free(F->child_ptr);
  free(F->child); free(F->rp); free(F->rows); free(F->lpos); free(F->amap_sn); free(F->amap_off); free(F->L); free(F->D);
  if (F->cb) { for (int64_t s = 0; s < F->ns; ++s) free(F->cb[s]); free(F->cb); }
  free(F->pending);
  free(F);
}

static int cmp_i64(const void *a, const void *b) {
  const int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

/* pattern analysis: perm may be NULL (natural order).  Only entries with row >= col are used. */
mf_factor *mf_analyze(int64_t n, const int64_t *colptr, const int64_t *rowval, int64_t base, const int64_t *perm) {
  mf_factor *F = (mf_factor *)calloc(1, sizeof(mf_factor));
  if (!F) return NULL;
  F->n = n;
  F->nnz_in = colptr[n] - base;
  F->perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
  F->iperm = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
  for (int64_t k = 0; k < n; ++k) F->perm[k] = perm ? perm[k] : k;
  for (int64_t k = 0; k < n; ++k) F->iperm[F->perm[k]] = k;
  /* permuted strictly-lower pattern by ROWS (for row i: columns j < i), duplicates allowed */
  int64_t *up = (int64_t *)calloc((size_t)(n + 2), sizeof(int64_t));
  for (int64_t j = 0; j < n; ++j)
    for (int64_t p = colptr[j] - base; p < colptr[j + 1] - base; ++p) {
      const int64_t i = rowval[p] - base;
      if (i <= j) continue;
      const int64_t a = F->iperm[i], b = F->iperm[j];
      ++up[(a > b ? a : b) + 1];
    }
  for (int64_t i = 0; i < n; ++i) up[i + 1] += up[i];
  int64_t *ui = (int64_t *)malloc(sizeof(int64_t) * (size_t)(up[n] + 1));
  {
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    memcpy(fill, up, sizeof(int64_t) * (size_t)n);
    for (int64_t j = 0; j < n; ++j)
      for (int64_t p = colptr[j] - base; p < colptr[j + 1] - base; ++p) {
        const int64_t i = rowval[p] - base;
        if (i <= j) continue;
        const int64_t a = F->iperm[i], b = F->iperm[j];
        ui[fill[a > b ? a : b]++] = a > b ? b : a;
      }
    free(fill);
  }
  /* elimination tree (Liu) and column counts by row-subtree traversal */
  F->parent = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
  F->count = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
  {
    int64_t *anc = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    for (int64_t i = 0; i < n; ++i) { F->parent[i] = -1; anc[i] = -1; }
    for (int64_t i = 0; i < n; ++i)
      for (int64_t p = up[i]; p < up[i + 1]; ++p) {
        int64_t k = ui[p];
        while (k != -1 && k < i) {
          const int64_t nxt = anc[k];
          anc[k] = i;
          if (nxt == -1) { F->parent[k] = i; break; }
          k = nxt;
        }
      }
    int64_t *mark = anc;
    for (int64_t i = 0; i < n; ++i) { F->count[i] = 1; mark[i] = -1; }
    for (int64_t i = 0; i < n; ++i) {
      mark[i] = i;
      for (int64_t p = up[i]; p < up[i + 1]; ++p)
        for (int64_t k = ui[p]; k != -1 && k < i && mark[k] != i; k = F->parent[k]) { ++F->count[k]; mark[k] = i; }
    }
    free(anc);
  }
  F->flops = 0;
  for (int64_t j = 0; j < n; ++j) F->flops += (double)F->count[j] * (double)F->count[j];
  /* fundamental supernodes: j joins j - 1 when parent[j - 1] == j, count[j - 1] == count[j] + 1 and j has one child */
  int64_t *nchild = (int64_t *)calloc((size_t)(n + 1), sizeof(int64_t));
  for (int64_t j = 0; j < n; ++j) if (F->parent[j] >= 0) ++nchild[F->parent[j]];
  F->sn0 = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 2));
  int64_t *col2sn = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
  F->ns = 0;
  for (int64_t j = 0; j < n; ++j) {
    const int join = j > 0 && F->parent[j - 1] == j && F->count[j - 1] == F->count[j] + 1 && nchild[j] == 1;
    if (!join) F->sn0[F->ns++] = j;
    col2sn[j] = F->ns - 1;
  }
  F->sn0[F->ns] = n;
  free(nchild);
  /* relaxed amalgamation (Ashcraft & Grimes 1989; CHOLMOD's nrelax rules are of the same kind): a supernode is merged into
   * the parent whose columns follow it directly when that costs few explicit zeros -- always up to 16 columns, up to 256
   * columns below 10 % zeros in the merged panel.  Without it the chains of one-column supernodes below a dense separator
   * re-add a whole contribution block per column. */
  int64_t *sf = (int64_t *)malloc(sizeof(int64_t) * (size_t)(F->ns + 1));     /* front order per (merged) supernode */
  {
    int64_t *m0 = (int64_t *)malloc(sizeof(int64_t) * (size_t)(F->ns + 1));
    int64_t *m1 = (int64_t *)malloc(sizeof(int64_t) * (size_t)(F->ns + 1));
    int64_t *mz = (int64_t *)malloc(sizeof(int64_t) * (size_t)(F->ns + 1));   /* explicit zeros already inside the merged panel */
    int64_t nm = 0;
    for (int64_t p = 0; p < F->ns; ++p) {
      int64_t c0 = F->sn0[p], c1 = F->sn0[p + 1], fp = F->count[c0], zp = 0;
      while (nm > 0) {
        const int64_t q = nm - 1;
        const int64_t pq = F->parent[m1[q] - 1];
        if (m1[q] != c0 || pq < c0 || pq >= c1) break;                         /* not the child that ends right before us */
        const int64_t kq = m1[q] - m0[q], kp = c1 - c0, fq = sf[q];
        const int64_t fnew = kq + fp;
        const int64_t zeros = zp + mz[q] + kq * (fnew - fq);
        const int64_t kn = kq + kp;
        const int64_t entries = fnew * kn - kn * (kn - 1) / 2;
        if (!(kn <= 16 || (kn <= 256 && zeros * 10 <= entries))) break;
        c0 = m0[q]; fp = fnew; zp = zeros;
        --nm;
      }
      m0[nm] = c0; m1[nm] = c1; sf[nm] = fp; mz[nm] = zp;
      ++nm;
    }
    for (int64_t t = 0; t < nm; ++t) { F->sn0[t] = m0[t]; for (int64_t j = m0[t]; j < m1[t]; ++j) col2sn[j] = t; }
    F->sn0[nm] = n;
    F->ns = nm;
    free(m0); free(m1); free(mz);
  }
  const int64_t ns = F->ns;
  F->sparent = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ns + 1));
  F->child_ptr = (int64_t *)calloc((size_t)(ns + 2), sizeof(int64_t));
  for (int64_t s = 0; s < ns; ++s) {
    const int64_t pj = F->parent[F->sn0[s + 1] - 1];
    F->sparent[s] = pj < 0 ? -1 : col2sn[pj];
    if (F->sparent[s] >= 0) ++F->child_ptr[F->sparent[s] + 1];
  }
  for (int64_t s = 0; s < ns; ++s) F->child_ptr[s + 1] += F->child_ptr[s];
  F->child = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ns + 1));
  {
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ns + 1));
    memcpy(fill, F->child_ptr, sizeof(int64_t) * (size_t)ns);
    for (int64_t s = 0; s < ns; ++s) if (F->sparent[s] >= 0) F->child[fill[F->sparent[s]]++] = s;
    free(fill);
  }
  /* front row lists: own columns, then the union of A's rows below and the children's rows beyond their own columns.
   * The supernodes are visited in column order; a child precedes its parent only when the permutation is a postorder
   * of the tree -- in general a parent can have a smaller index than a child's descendant... it cannot: parent[j] > j. */
  F->rp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ns + 2));
  F->lpos = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ns + 2));
  F->rp[0] = 0; F->lpos[0] = 0;
  for (int64_t s = 0; s < ns; ++s) {
    const int64_t f = sf[s], k = F->sn0[s + 1] - F->sn0[s];
    F->rp[s + 1] = F->rp[s] + f;
    F->lpos[s + 1] = F->lpos[s] + f * k;
    if (f > F->max_front) F->max_front = f;
  }
  free(sf);
  F->rows = (int64_t *)malloc(sizeof(int64_t) * (size_t)(F->rp[ns] + 1));
  /* column lists of the permuted strictly-lower pattern (rows > col) */
  int64_t *lp = (int64_t *)calloc((size_t)(n + 2), sizeof(int64_t));
  for (int64_t i = 0; i < n; ++i) for (int64_t p = up[i]; p < up[i + 1]; ++p) ++lp[ui[p] + 1];
  for (int64_t j = 0; j < n; ++j) lp[j + 1] += lp[j];
  int64_t *li = (int64_t *)malloc(sizeof(int64_t) * (size_t)(lp[n] + 1));
  {
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    memcpy(fill, lp, sizeof(int64_t) * (size_t)n);
    for (int64_t i = 0; i < n; ++i) for (int64_t p = up[i]; p < up[i + 1]; ++p) li[fill[ui[p]]++] = i;
    free(fill);
  }
  {
    int64_t *mark = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    for (int64_t i = 0; i < n; ++i) mark[i] = -1;
    int ok = 1;
    for (int64_t s = 0; s < ns && ok; ++s) {
      const int64_t c0 = F->sn0[s], c1 = F->sn0[s + 1], f = F->rp[s + 1] - F->rp[s];
      int64_t *R = F->rows + F->rp[s];
      int64_t cnt = 0;
      for (int64_t j = c0; j < c1; ++j) { R[cnt++] = j; mark[j] = s; }
      for (int64_t j = c0; j < c1; ++j)
        for (int64_t p = lp[j]; p < lp[j + 1]; ++p) {
          const int64_t i = li[p];
          if (mark[i] != s) { mark[i] = s; if (cnt < f) R[cnt] = i; ++cnt; }
        }
      for (int64_t q = F->child_ptr[s]; q < F->child_ptr[s + 1]; ++q) {
        const int64_t c = F->child[q];
        const int64_t kc = F->sn0[c + 1] - F->sn0[c];
        for (int64_t t = F->rp[c] + kc; t < F->rp[c + 1]; ++t) {
          const int64_t i = F->rows[t];
          if (mark[i] != s) { mark[i] = s; if (cnt < f) R[cnt] = i; ++cnt; }
        }
      }
      if (cnt != f) ok = 0;      /* structure and column count disagree: internal error */
      qsort(R + (c1 - c0), (size_t)(f - (c1 - c0)), sizeof(int64_t), cmp_i64);
    }
    free(mark);
    if (!ok) { free(up); free(ui); free(lp); free(li); free(col2sn); mf_free(F); return NULL; }
  }
  /* scatter map of the input entries */
  F->amap_sn = (int64_t *)malloc(sizeof(int64_t) * (size_t)(F->nnz_in + 1));
  F->amap_off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(F->nnz_in + 1));
  {
    int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    /* per supernode positions are needed per entry: entries are grouped by their (permuted) column's supernode */
    for (int64_t j = 0; j < n; ++j)
      for (int64_t p = colptr[j] - base; p < colptr[j + 1] - base; ++p) F->amap_sn[p] = -1;
    /* bucket entries by supernode to reuse one position map */
    int64_t *cnt = (int64_t *)calloc((size_t)(ns + 2), sizeof(int64_t));
    for (int64_t j = 0; j < n; ++j)
      for (int64_t p = colptr[j] - base; p < colptr[j + 1] - base; ++p) {
        const int64_t i = rowval[p] - base;
        if (i < j) continue;
        const int64_t a = F->iperm[i], b = F->iperm[j];
        ++cnt[col2sn[a < b ? a : b] + 1];
      }
    for (int64_t s = 0; s < ns; ++s) cnt[s + 1] += cnt[s];
    int64_t *ent = (int64_t *)malloc(sizeof(int64_t) * (size_t)(cnt[ns] + 1));
    int64_t *er = (int64_t *)malloc(sizeof(int64_t) * (size_t)(cnt[ns] + 1));
    int64_t *ec = (int64_t *)malloc(sizeof(int64_t) * (size_t)(cnt[ns] + 1));
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ns + 1));
    memcpy(fill, cnt, sizeof(int64_t) * (size_t)ns);
    for (int64_t j = 0; j < n; ++j)
      for (int64_t p = colptr[j] - base; p < colptr[j + 1] - base; ++p) {
        const int64_t i = rowval[p] - base;
        if (i < j) continue;
        const int64_t a = F->iperm[i], b = F->iperm[j];
        const int64_t c = a < b ? a : b, r = a < b ? b : a;
        const int64_t q = fill[col2sn[c]]++;
        ent[q] = p; er[q] = r; ec[q] = c;
      }
    for (int64_t s = 0; s < ns; ++s) {
      const int64_t f = F->rp[s + 1] - F->rp[s];
      for (int64_t t = 0; t < f; ++t) pos[F->rows[F->rp[s] + t]] = t;
      for (int64_t q = cnt[s]; q < cnt[s + 1]; ++q) {
        F->amap_sn[ent[q]] = s;
        F->amap_off[ent[q]] = pos[er[q]] + (ec[q] - F->sn0[s]) * f;
      }
    }
    free(pos); free(cnt); free(ent); free(er); free(ec); free(fill);
  }
  F->L = (double *)malloc(sizeof(double) * (size_t)(F->lpos[ns] + 1));
  F->D = (double *)malloc(sizeof(double) * (size_t)(n + 1));
  F->cb = (double **)calloc((size_t)(ns + 1), sizeof(double *));
  F->pending = (int *)malloc(sizeof(int) * (size_t)(ns + 1));
  free(up); free(ui); free(lp); free(li); free(col2sn);
  if (!F->L || !F->D || !F->cb || !F->pending) { mf_free(F); return NULL; }
  return F;
}

double mf_flops(const mf_factor *F) { return F->flops; }
int64_t mf_nsuper(const mf_factor *F) { return F->ns; }
int64_t mf_max_front(const mf_factor *F) { return F->max_front; }
int64_t mf_lnz(const mf_factor *F) { return F->lpos[F->ns]; }
void mf_get_D(const mf_factor *F, double *out) { memcpy(out, F->D, sizeof(double) * (size_t)F->n); }

/* ---- dense kernels on a column-major lower-triangular front ------------------------------------------------------- */
#define MF_NB 64
/* C[i, c] -= sum_p W[i, p] * Lc[c, p] for a tile: rows [i0, i1), columns [c0, c1), only i >= c (column-major, ld) */
static void tile_update(double *restrict A, int64_t ld, const double *restrict W, int64_t ldw, int64_t j0, int64_t jb,
                        int64_t i0, int64_t i1, int64_t c0, int64_t c1) {
  for (int64_t c = c0; c < c1; ++c) {
    const int64_t is = i0 > c ? i0 : c;
    double *restrict Ac = A + c * ld;
    for (int64_t p = 0; p < jb; ++p) {
      const double t = A[(j0 + p) * ld + c];          /* L[c, j0 + p] (already divided by d) */
      const double *restrict Wp = W + p * ldw;
      for (int64_t i = is; i < i1; ++i) Ac[i] -= Wp[i] * t;
    }
  }
}

/* rows [i0, i1) below the diagonal block of the panel [j0, j0 + jb): W = A21 L11^-T (right-looking over the panel's columns,
 * L11 and D final in the diagonal block), L21 = W D^-1; independent across rows */
static void panel_rows(double *restrict A, int64_t f, double *restrict W, int64_t j0, int64_t jb, int64_t i0, int64_t i1) {
  for (int64_t j = j0; j < j0 + jb; ++j) {
    const double rd = 1.0 / A[j * f + j];
    double *restrict Aj = A + j * f;
    double *restrict Wj = W + (j - j0) * f;
    const int64_t is = i0 > j + 1 ? i0 : j + 1;  /* strictly below the pivot (matters inside the diagonal block only) */
    for (int64_t i = is; i < i1; ++i) { Wj[i] = Aj[i]; Aj[i] *= rd; }
    for (int64_t c = j + 1; c < j0 + jb; ++c) {
      const double t = Aj[c];                    /* L[c, j] of the diagonal block */
      double *restrict Ac = A + c * f;
      const int64_t ic = is > c ? is : c;        /* lower triangle only */
      for (int64_t i = ic; i < i1; ++i) Ac[i] -= Wj[i] * t;
    }
  }
}

/* partial LDL^T of the leading k columns of the f x f front A (lower), blocked; W: scratch f * MF_NB */
static void front_factor(double *A, int64_t f, int64_t k, double *W, int par) {
  for (int64_t j0 = 0; j0 < k; j0 += MF_NB) {
    const int64_t jb = (k - j0) < MF_NB ? (k - j0) : MF_NB;
    const int64_t t0 = j0 + jb;
    /* the jb x jb diagonal block, unblocked (the only serial piece of a panel) */
    panel_rows(A, f, W, j0, jb, j0, t0);          /* rows inside the block: entries above the diagonal of a column are never read */
    if (t0 >= f) break;
    /* the rows below, then the trailing update: tiles of MF_NB columns x 256 rows */
    if (par && f - t0 > 1024) {
      const int64_t nrt = (f - t0 + 255) / 256;
#pragma omp taskloop grainsize(1) default(shared)
      for (int64_t rt = 0; rt < nrt; ++rt) {
        const int64_t i0 = t0 + rt * 256;
        panel_rows(A, f, W, j0, jb, i0, (i0 + 256) < f ? (i0 + 256) : f);
      }
      const int64_t nct = (f - t0 + MF_NB - 1) / MF_NB;
#pragma omp taskloop grainsize(1) default(shared)
      for (int64_t ct = 0; ct < nct; ++ct) {
        const int64_t c0 = t0 + ct * MF_NB, c1 = (c0 + MF_NB) < f ? (c0 + MF_NB) : f;
        for (int64_t i0 = c0; i0 < f; i0 += 256) tile_update(A, f, W, f, j0, jb, i0, (i0 + 256) < f ? (i0 + 256) : f, c0, c1);
      }
    } else {
      panel_rows(A, f, W, j0, jb, t0, f);
      for (int64_t c0 = t0; c0 < f; c0 += MF_NB) {
        const int64_t c1 = (c0 + MF_NB) < f ? (c0 + MF_NB) : f;
        for (int64_t i0 = c0; i0 < f; i0 += 256) tile_update(A, f, W, f, j0, jb, i0, (i0 + 256) < f ? (i0 + 256) : f, c0, c1);
      }
    }
  }
}

static void do_front(mf_factor *F, const double *vals, const int64_t *ent_ptr, const int64_t *ent, int64_t s, int par) {
  const int64_t c0 = F->sn0[s], k = F->sn0[s + 1] - c0, f = F->rp[s + 1] - F->rp[s], r = f - k;
  const int64_t *R = F->rows + F->rp[s];
  double *A = (double *)calloc((size_t)(f * f + 1), sizeof(double));
  double *W = (double *)malloc(sizeof(double) * (size_t)(f * MF_NB + 1));
  for (int64_t q = ent_ptr[s]; q < ent_ptr[s + 1]; ++q) A[F->amap_off[ent[q]]] += vals[ent[q]];
  /* extend-add of the children, in child order (deterministic) */
  for (int64_t q = F->child_ptr[s]; q < F->child_ptr[s + 1]; ++q) {
    const int64_t c = F->child[q];
    const int64_t kc = F->sn0[c + 1] - F->sn0[c], fc = F->rp[c + 1] - F->rp[c], rc = fc - kc;
    const int64_t *Rc = F->rows + F->rp[c] + kc;
    int64_t *rel = (int64_t *)malloc(sizeof(int64_t) * (size_t)(rc + 1));
    for (int64_t t = 0, u = 0; t < rc; ++t) { while (R[u] != Rc[t]) ++u; rel[t] = u; }      /* both lists are sorted */
    const double *C = F->cb[c] + kc * fc + kc;        /* the child's front buffer: trailing rc x rc block, leading dimension fc */
    for (int64_t jj = 0; jj < rc; ++jj) {
      double *Ac = A + rel[jj] * f;
      const double *Cc = C + jj * fc;
      for (int64_t ii = jj; ii < rc; ++ii) Ac[rel[ii]] += Cc[ii];
    }
    free(rel);
    free(F->cb[c]);
    F->cb[c] = NULL;
  }
  front_factor(A, f, k, W, par);
  /* keep the panel and D, pass the contribution block on */
  double *Lp = F->L + F->lpos[s];
  for (int64_t j = 0; j < k; ++j) {
    F->D[c0 + j] = A[j * f + j];
    memcpy(Lp + j * f, A + j * f, sizeof(double) * (size_t)f);
  }
  if (r > 0) F->cb[s] = A;      /* the contribution block stays where it is until the parent has added it */
  else free(A);
  free(W);
}

/* the whole subtree below s (s excluded), children before parents, on this thread: an explicit stack instead of recursion
 * (a banded matrix has an elimination tree of depth n) */
static void run_subtree_seq(mf_factor *F, const double *vals, const int64_t *ent_ptr, const int64_t *ent, int64_t root) {
  int64_t cap = 64, top = 0;
  int64_t *stk = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap * 2);
  stk[0] = root; stk[1] = F->child_ptr[root]; top = 1;
  while (top > 0) {
    const int64_t s = stk[2 * (top - 1)];
    int64_t *it = &stk[2 * (top - 1) + 1];
    if (*it < F->child_ptr[s + 1]) {
      const int64_t c = F->child[(*it)++];
      if (top == cap) { cap *= 2; stk = (int64_t *)realloc(stk, sizeof(int64_t) * (size_t)cap * 2); }
      stk[2 * top] = c; stk[2 * top + 1] = F->child_ptr[c]; ++top;
    } else {
      if (s != root) do_front(F, vals, ent_ptr, ent, s, 0);
      --top;
    }
  }
  free(stk);
}

static void run_front_task(mf_factor *F, const double *vals, const int64_t *ent_ptr, const int64_t *ent, int64_t s, int whole_subtree) {
  /* (the subtree below s, then) the front s, then up the tree while this task is the last child to finish */
  if (whole_subtree) run_subtree_seq(F, vals, ent_ptr, ent, s);
  for (;;) {
    const int64_t f = F->rp[s + 1] - F->rp[s];
    do_front(F, vals, ent_ptr, ent, s, f > 1500);
    const int64_t p = F->sparent[s];
    if (p < 0) return;
    int left;
#pragma omp atomic capture
    left = --F->pending[p];
    if (left != 0) return;
    s = p;
  }
}

/* numeric factorisation with `nthreads` OpenMP threads (<= 0: the runtime's default).  sym_kind 0: Cholesky semantics
 * (success <=> all pivots > 0), 1: LDL^T, inertia must be (npos, nneg, 0) with tolerance tol.  Returns 1 / 0; counts out. */
int mf_factor_numeric(mf_factor *F, const double *vals, int64_t npos, int64_t nneg, int sym_kind, double tol, int nthreads, int64_t counts[4]) {
  const int64_t ns = F->ns;
  /* entries grouped by supernode */
  int64_t *ent_ptr = (int64_t *)calloc((size_t)(ns + 2), sizeof(int64_t));
  for (int64_t p = 0; p < F->nnz_in; ++p) if (F->amap_sn[p] >= 0) ++ent_ptr[F->amap_sn[p] + 1];
  for (int64_t s = 0; s < ns; ++s) ent_ptr[s + 1] += ent_ptr[s];
  int64_t *ent = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ent_ptr[ns] + 1));
  {
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ns + 1));
    memcpy(fill, ent_ptr, sizeof(int64_t) * (size_t)ns);
    for (int64_t p = 0; p < F->nnz_in; ++p) if (F->amap_sn[p] >= 0) ent[fill[F->amap_sn[p]]++] = p;
    free(fill);
  }
  for (int64_t s = 0; s < ns; ++s) F->pending[s] = (int)(F->child_ptr[s + 1] - F->child_ptr[s]);
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
  const int nth = nthreads > 0 ? nthreads : omp_get_max_threads();
#else
  const int nth = 1;
#endif
  /* One task per maximal subtree of little work (run on one thread, children first) and one per front above those: tens of
   * thousands of leaf-sized tasks would only fight over the task queue.  Dense flops of a front ~ k f^2. */
  double *sub = (double *)malloc(sizeof(double) * (size_t)(ns + 1));
  double total = 0.0;
  for (int64_t s = 0; s < ns; ++s) {
    const double f = (double)(F->rp[s + 1] - F->rp[s]), k = (double)(F->sn0[s + 1] - F->sn0[s]);
    sub[s] = k * f * f + 1e3;
  }
  for (int64_t s = 0; s < ns; ++s) { total += sub[s]; if (F->sparent[s] >= 0) sub[F->sparent[s]] += sub[s]; }
  const double small = total / (16.0 * (double)nth) > 2e7 ? total / (16.0 * (double)nth) : 2e7;
#pragma omp parallel default(shared)
  {
#pragma omp single
    {
      for (int64_t s = 0; s < ns; ++s) {
        const int64_t p = F->sparent[s];
        const int is_small = sub[s] <= small;
        const int parent_small = p >= 0 && sub[p] <= small;
        if (parent_small) continue;                        /* inside somebody else's sequential subtree */
        if (is_small) {
#pragma omp task firstprivate(s) default(shared)
          run_front_task(F, vals, ent_ptr, ent, s, 1);
        } else if (F->child_ptr[s + 1] == F->child_ptr[s]) {
#pragma omp task firstprivate(s) default(shared)
          run_front_task(F, vals, ent_ptr, ent, s, 0);
        }
      }
    }
  }
  free(sub);
  free(ent_ptr);
  free(ent);
  int64_t pos = 0, neg = 0, zer = 0, bad = 0;
  const double t = sym_kind == 0 ? 0.0 : tol;
  for (int64_t j = 0; j < F->n; ++j) {
    const double d = F->D[j];
    if (isnan(d) || isinf(d)) ++bad;
    else if (d > t) ++pos;
    else if (d < -t) ++neg;
    else ++zer;
  }
  if (counts) { counts[0] = pos; counts[1] = neg; counts[2] = zer; counts[3] = bad; }
  if (bad > 0) return 0;
  return sym_kind == 0 ? (pos == F->n) : (pos == npos && neg == nneg);
}

/* sol = P' L^-T D^-1 L^-1 P rhs, supernode by supernode (one thread) */
void mf_solve(const mf_factor *F, const double *rhs, double *sol) {
  const int64_t n = F->n, ns = F->ns;
  double *x = (double *)malloc(sizeof(double) * (size_t)(n + 1));
  for (int64_t k = 0; k < n; ++k) x[k] = rhs[F->perm[k]];
  for (int64_t s = 0; s < ns; ++s) {
    const int64_t c0 = F->sn0[s], k = F->sn0[s + 1] - c0, f = F->rp[s + 1] - F->rp[s];
    const int64_t *R = F->rows + F->rp[s];
    const double *Lp = F->L + F->lpos[s];
    for (int64_t j = 0; j < k; ++j) {
      const double xj = x[c0 + j];
      const double *Lj = Lp + j * f;
      for (int64_t i = j + 1; i < f; ++i) x[R[i]] -= Lj[i] * xj;
    }
  }
  for (int64_t j = 0; j < n; ++j) x[j] /= F->D[j];
  for (int64_t s = ns - 1; s >= 0; --s) {
    const int64_t c0 = F->sn0[s], k = F->sn0[s + 1] - c0, f = F->rp[s + 1] - F->rp[s];
    const int64_t *R = F->rows + F->rp[s];
    const double *Lp = F->L + F->lpos[s];
    for (int64_t j = k - 1; j >= 0; --j) {
      const double *Lj = Lp + j * f;
      double acc = x[c0 + j];
      for (int64_t i = j + 1; i < f; ++i) acc -= Lj[i] * x[R[i]];
      x[c0 + j] = acc;
    }
  }
  for (int64_t k = 0; k < n; ++k) sol[F->perm[k]] = x[k];
  free(x);
}
