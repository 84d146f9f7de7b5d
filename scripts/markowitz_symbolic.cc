// Study aid (CPU): sequential minimum-Markowitz elimination of a sparse PATTERN (no values, no threshold test) -- the fill a sparse LU of
// that pattern cannot avoid by ordering alone.  Input: int32 n, int64 nnz, int32 rows[nnz], int32 cols[nnz].  Prints nnz(L)+nnz(U) and the
// point where the active submatrix passes 30 % density.  g++ -O2 -o markowitz_symbolic scripts/markowitz_symbolic.cc
// Used for profiles/r05_lu_fill_study.txt (scripts/lu_fill_study.py).
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef uint64_t u64;
int main(int argc, char** argv) {
    int n; long nnz;
    FILE* f = fopen(argv[1], "rb");
    fread(&n, 4, 1, f); fread(&nnz, 8, 1, f);
    std::vector<int> I(nnz), J(nnz);
    fread(I.data(), 4, nnz, f); fread(J.data(), 4, nnz, f);
    int W = (n + 63) / 64;
    std::vector<u64> R((size_t)n * W, 0);   // row bitsets over columns
    std::vector<int> rc(n, 0), cc(n, 0);
    for (long k = 0; k < nnz; k++) { R[(size_t)I[k] * W + J[k] / 64] |= 1ull << (J[k] % 64); rc[I[k]]++; cc[J[k]]++; }
    std::vector<char> ra(n, 1), ca(n, 1);
    long lu = 0;
    int dense_at = -1;
    for (int step = 0; step < n; step++) {
        // min Markowitz: search rows by count ascending (cheap approx: find min over all active entries for rows with min few counts)
        long best = -1; int bi = -1, bj = -1;
        // candidate rows: those with smallest rc (up to 4 distinct smallest rows), candidate columns likewise
        int minr = 1 << 30;
        for (int i = 0; i < n; i++) if (ra[i] && rc[i] < minr) minr = rc[i];
        for (int i = 0; i < n; i++) if (ra[i] && rc[i] <= minr + 1) {
            for (int w = 0; w < W; w++) { u64 b = R[(size_t)i * W + w]; while (b) { int j = w * 64 + __builtin_ctzll(b); b &= b - 1;
                long cost = (long)(rc[i] - 1) * (cc[j] - 1); if (best < 0 || cost < best) { best = cost; bi = i; bj = j; } } }
            if (best == 0) break;
        }
        int minc = 1 << 30, jc = -1;
        for (int j = 0; j < n; j++) if (ca[j] && cc[j] < minc) { minc = cc[j]; jc = j; }
        if (best != 0 && jc >= 0) {
            for (int i = 0; i < n; i++) if (ra[i] && (R[(size_t)i * W + jc / 64] >> (jc % 64) & 1)) {
                long cost = (long)(rc[i] - 1) * (cc[jc] - 1); if (best < 0 || cost < best) { best = cost; bi = i; bj = jc; } }
        }
        // eliminate pivot (bi, bj)
        lu += rc[bi] + cc[bj] - 1;
        long active = n - step;
        ra[bi] = 0; ca[bj] = 0;
        u64* pr = &R[(size_t)bi * W];
        for (int w = 0; w < W; w++) { u64 b = pr[w]; while (b) { int j = w * 64 + __builtin_ctzll(b); b &= b - 1; cc[j]--; } }
        pr[bj / 64] &= ~(1ull << (bj % 64));
        for (int i = 0; i < n; i++) if (ra[i] && (R[(size_t)i * W + bj / 64] >> (bj % 64) & 1)) {
            u64* r = &R[(size_t)i * W];
            r[bj / 64] &= ~(1ull << (bj % 64));
            int cnt = 0;
            for (int w = 0; w < W; w++) { u64 nw = pr[w] & ~r[w]; while (nw) { int j = w * 64 + __builtin_ctzll(nw); nw &= nw - 1; cc[j]++; } r[w] |= pr[w]; cnt += __builtin_popcountll(r[w]); }
            rc[i] = cnt;
        }
        if (dense_at < 0) { // density of the active submatrix
            if (step % 256 == 0) { long tot = 0; for (int i = 0; i < n; i++) if (ra[i]) tot += rc[i]; double dens = (double)tot / ((double)(active - 1) * (active - 1) + 1);
                if (step % 1024 == 0) fprintf(stderr, "step %d active %ld entries %ld density %.3f lu so far %ld\n", step, active - 1, tot, dens, lu);
                if (dens > 0.3) { dense_at = step; fprintf(stderr, "dense (>30%%) at step %d: remaining %ld, lu so far %ld\n", step, active - 1, lu); } }
        }
    }
    printf("n %d nnz %ld  nnz(L+U) %ld ratio %.1f\n", n, nnz, lu, (double)lu / nnz);
}
