// sched_sim.cpp -- CPU model of the LDS-ring iteration scheduler with bank-class caps (design probe
// for round 3; not part of the product).  One consumer wave's stream at config-4 statistics: ROWS
// rows, ~PER_CHUNK half-edges per 1024-column chunk, NCHUNK chunks.  An iteration takes up to 64
// entries whose chunks lie within SPAN of the oldest pending one; rows are distinct inside an
// iteration (mandatory); at most capR entries per row bank class (row mod 32) and capC per column
// bank class (col mod 32) unless the entry has been deferred `force` times already (soft caps).
//   g++ -O2 -o /tmp/sched_sim tools/sched_sim.cpp && /tmp/sched_sim
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "ring_place_r3.h"

struct Ent { int chunk, row, col, age; };

struct Res { double iters, padpct, Kr, Kc, rd_r, rd_c, wr, forced, xr, xc, xw; };

static int g_adjacent = 0;  // 1: a row taken in iteration k is not taken in k + 1 either
static Res run(int ROWS, double PER_CHUNK, int NCHUNK, int SPAN, int capR, int capC, int force, int CARRY, unsigned seed) {
  std::mt19937_64 rng(seed);
  std::poisson_distribution<int> pois(PER_CHUNK);
  std::vector<Ent> st;
  for (int j = 0; j < NCHUNK; ++j) {
    int k = pois(rng);
    for (int t = 0; t < k; ++t) st.push_back({j, (int)(rng() % ROWS), (int)(rng() % 1024), 0});
  }
  // CSR order inside a chunk is by row (stable sort by chunk of a row-major list)
  std::stable_sort(st.begin(), st.end(), [](const Ent& a, const Ent& b) { return a.chunk != b.chunk ? a.chunk < b.chunk : a.row < b.row; });
  // row -> bank class: rows dealt to classes round-robin (slot = local row index)
  std::vector<Ent> q;  // deferred, FIFO
  size_t pos = 0;
  long x_rr = 0, x_rc = 0, x_w = 0; long iters = 0, pad = 0, sKr = 0, sKc = 0, srr = 0, src = 0, swr = 0, nforced = 0;
  std::vector<int> rowtaken(ROWS, -5);
  while (pos < st.size() || !q.empty()) {
    const int m = !q.empty() ? std::min(q[0].chunk, pos < st.size() ? st[pos].chunk : 1 << 30) : st[pos].chunk;
    int mm = m;
    for (auto& e : q) mm = std::min(mm, e.chunk);
    const int lim = mm + SPAN;
    int rc[32] = {0}, cc[32] = {0}, rc16[16] = {0};
    std::vector<Ent> take, keep;
    auto offer = [&](Ent e) {
      const int a = e.row & 31, b = e.col & 31;
      const bool rowfree = rowtaken[e.row] != (int)iters && !(g_adjacent && rowtaken[e.row] == (int)iters - 1);
      const bool capok = (rc[a] < capR && cc[b] < capC) || e.age >= force;
      if (take.size() < 64 && rowfree && capok) {
        if (!(rc[a] < capR && cc[b] < capC)) ++nforced;
        take.push_back(e); rowtaken[e.row] = (int)iters; ++rc[a]; ++cc[b]; ++rc16[a & 15];
      } else {
        if (rowfree || take.size() >= 64) {} // age only counts real rejections
        e.age += (take.size() < 64) ? 1 : 0;
        keep.push_back(e);
      }
    };
    for (auto& e : q) offer(e);
    while (take.size() < 64 && pos < st.size() && st[pos].chunk <= lim && keep.size() + 1 <= (size_t)CARRY) offer(st[pos++]);
    q.swap(keep);
    ++iters;
    pad += 64 - (long)take.size();
    {
      uint8_t rcl[64], ccl[64], lane_of[64], fl[64];
      static RingPlaceScratch S;
      const int cnt = (int)take.size();
      for (int e = 0; e < cnt; ++e) { rcl[e] = take[e].row & 31; ccl[e] = take[e].col & 31; }
      ring_place(cnt, rcl, ccl, lane_of, fl, S);
      int hr[2][32] = {{0}}, hc[2][32] = {{0}}, qw[4][16] = {{0}};
      bool seen[64] = {false};
      for (int e = 0; e < cnt; ++e) {
        const int l = lane_of[e];
        if (l > 63 || seen[l]) { printf("placement error\n"); exit(1); }
        seen[l] = true;
        ++hr[l >> 5][rcl[e]]; ++hc[l >> 5][ccl[e]]; ++qw[l >> 4][rcl[e] & 15];
      }
      int cr = 0, cc2 = 0, cw = 0;
      for (int h = 0; h < 2; ++h) { int a = 1, b = 1; for (int c = 0; c < 32; ++c) { a = std::max(a, hr[h][c]); b = std::max(b, hc[h][c]); } cr += a; cc2 += b; }
      for (int q4 = 0; q4 < 4; ++q4) { int a = 1; for (int c = 0; c < 16; ++c) a = std::max(a, qw[q4][c]); cw += a; }
      x_rr += cr; x_rc += cc2; x_w += std::max(6, cw);
    }
    int Kr = 0, Kc = 0, K16 = 0;
    for (int c = 0; c < 32; ++c) { Kr = std::max(Kr, rc[c]); Kc = std::max(Kc, cc[c]); }
    for (int c = 0; c < 16; ++c) K16 = std::max(K16, rc16[c]);
    sKr += Kr; sKc += Kc;
    srr += std::max(2, Kr); src += std::max(2, Kc);           // two 32-lane passes, classes split evenly
    swr += std::max(6, 4 * ((K16 + 3) / 4));
  }
  Res r;
  r.iters = (double)iters; r.padpct = 100.0 * pad / (64.0 * iters);
  r.Kr = (double)sKr / iters; r.Kc = (double)sKc / iters;
  r.rd_r = (double)srr / iters; r.rd_c = (double)src / iters; r.wr = (double)swr / iters;
  r.forced = (double)nforced / iters;
  r.xr = (double)x_rr / iters; r.xc = (double)x_rc / iters; r.xw = (double)x_w / iters;
  return r;
}

int main(int argc, char** argv) {
  const int ROWS = argc > 1 ? atoi(argv[1]) : 326;
  const double PER = argc > 2 ? atof(argv[2]) : 32.0;
  g_adjacent = argc > 3 ? atoi(argv[3]) : 0;
  const int NCH = 977;
  printf("rows/wave %d, %.0f entries per chunk\n", ROWS, PER);
  printf("span capR capC force | iters  pad%%   Krow Kcol | LDS clk: 2 row reads + col read + write + codebook = total | x(1+pad) | forced/iter\n");
  const int caps[][2] = {{64, 64}, {4, 4}};
  for (int span : {4, 5, 6})
    for (auto& c : caps)
      for (int force : {2, 3, 1000}) {
        if (c[0] == 64 && force != 2) continue;
        Res a = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const int T = 6;
        for (int s = 0; s < T; ++s) {
          Res r = run(ROWS, PER, NCH, span, c[0], c[1], force, 192, 1234 + s);
          a.iters += r.iters / T; a.padpct += r.padpct / T; a.Kr += r.Kr / T; a.Kc += r.Kc / T;
          a.rd_r += r.rd_r / T; a.rd_c += r.rd_c / T; a.wr += r.wr / T; a.forced += r.forced / T; a.xr += r.xr / T; a.xc += r.xc / T; a.xw += r.xw / T;
        }
        const double tot = 2 * a.rd_r + a.rd_c + a.wr + 2.0;
        const double xt = 2 * a.xr + a.xc + a.xw + 2.4;
        printf("%4d %4d %4d %5d | %5.0f %5.1f  %4.2f %4.2f | %5.2f %5.2f %5.2f -> %5.1f | %5.1f | %.2f || placed: %5.2f %5.2f %5.2f -> %5.1f x(1+pad) %5.1f\n", span, c[0], c[1], force, a.iters,
               a.padpct, a.Kr, a.Kc, 2 * a.rd_r, a.rd_c, a.wr, tot, tot * (1.0 + a.padpct / 100.0), a.forced, 2 * a.xr, a.xc, a.xw, xt, xt * (1.0 + a.padpct / 100.0));
      }
  return 0;
}
