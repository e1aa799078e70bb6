// Rectangular linear sum assignment on the host (the matcher's combinatorial step).
// reference call sites: models/detr/matcher.py:80, models/detr/matcher_ucf.py:82 ->
// scipy.optimize.linear_sum_assignment (third-party, not vendored in the reference; SciPy is unpinned there).
// This restates the published algorithm SciPy implements -- the shortest-augmenting-path method of
// D. F. Crouse, "On implementing 2D rectangular assignment algorithms", IEEE TAES 52(4), 2016 --
// including its tie-breaking (prefer an unassigned column among equal shortest paths; scan the
// remaining columns in reverse index order), so that cost ties -- which DO occur for AVA because the
// class cost is constant across targets (matcher.py:72) -- resolve exactly as in the reference.
// The cost matrices are tiny (15 x N_i, or 10 x 1), so this runs on the CPU between two kernel launches.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <numeric>
#include <vector>

namespace {

int augmenting_path(int nc, const double* cost, std::vector<double>& u, std::vector<double>& v, std::vector<int>& path,
                    std::vector<int>& row4col, std::vector<double>& spc, int i, std::vector<char>& SR, std::vector<char>& SC,
                    std::vector<int>& remaining, double* p_min) {
    double min_val = 0;
    int num_remaining = nc;
    for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
    std::fill(SR.begin(), SR.end(), 0);
    std::fill(SC.begin(), SC.end(), 0);
    std::fill(spc.begin(), spc.end(), std::numeric_limits<double>::infinity());
    int sink = -1;
    while (sink == -1) {
        int index = -1;
        double lowest = std::numeric_limits<double>::infinity();
        SR[i] = 1;
        for (int it = 0; it < num_remaining; ++it) {
            const int j = remaining[it];
            const double r = min_val + cost[(long)i * nc + j] - u[i] - v[j];
            if (r < spc[j]) { path[j] = i; spc[j] = r; }
            if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
        }
        min_val = lowest;
        if (min_val == std::numeric_limits<double>::infinity()) return -1;
        const int j = remaining[index];
        if (row4col[j] == -1) sink = j; else i = row4col[j];
        SC[j] = 1;
        remaining[index] = remaining[--num_remaining];
    }
    *p_min = min_val;
    return sink;
}

}  // namespace

extern "C" {

// cost: row-major [nr][nc] doubles (host).  Writes min(nr,nc) pairs (row_ind[k], col_ind[k]) sorted by row,
// as scipy.optimize.linear_sum_assignment returns them.  Returns 0, -1 on bad arguments, -2 if infeasible.
int tuber_lsap(const double* cost_in, int nr, int nc, long* row_ind, long* col_ind) {
    if (nr < 0 || nc < 0) return -1;
    if (nr == 0 || nc == 0) return 0;
    const bool transpose = nc < nr;
    std::vector<double> cost((size_t)nr * nc);
    if (transpose) {
        for (int i = 0; i < nr; ++i)
            for (int j = 0; j < nc; ++j) cost[(size_t)j * nr + i] = cost_in[(size_t)i * nc + j];
        std::swap(nr, nc);
    } else {
        std::copy(cost_in, cost_in + (size_t)nr * nc, cost.begin());
    }
    for (double c : cost)
        if (std::isnan(c) || c == -std::numeric_limits<double>::infinity()) return -1;
    std::vector<double> u(nr, 0), v(nc, 0), spc(nc);
    std::vector<int> path(nc, -1), col4row(nr, -1), row4col(nc, -1), remaining(nc);
    std::vector<char> SR(nr), SC(nc);
    for (int cur = 0; cur < nr; ++cur) {
        double min_val;
        const int sink = augmenting_path(nc, cost.data(), u, v, path, row4col, spc, cur, SR, SC, remaining, &min_val);
        if (sink < 0) return -2;
        u[cur] += min_val;
        for (int i = 0; i < nr; ++i)
            if (SR[i] && i != cur) u[i] += min_val - spc[col4row[i]];
        for (int j = 0; j < nc; ++j)
            if (SC[j]) v[j] -= min_val - spc[j];
        int j = sink;
        while (true) {
            const int i = path[j];
            row4col[j] = i;
            std::swap(col4row[i], j);
            if (i == cur) break;
        }
    }
    if (transpose) {
        std::vector<int> order(nr);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return col4row[a] < col4row[b]; });
        for (int k = 0; k < nr; ++k) { row_ind[k] = col4row[order[k]]; col_ind[k] = order[k]; }
    } else {
        for (int i = 0; i < nr; ++i) { row_ind[i] = i; col_ind[i] = col4row[i]; }
    }
    return 0;
}

}  // extern "C"
