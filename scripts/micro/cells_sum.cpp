#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int SUB_NONE = 1 << 29;
inline int h_rng_lo(int pl, int tl, int sub, int s) { return std::max(std::max(-pl, -s), (tl - pl) - sub + s); }
inline int h_rng_hi(int pl, int tl, int sub, int s) { return std::min(std::min(tl, s), (tl - pl) + sub - s); }
inline int64_t h_row_cells(int pl, int tl, int sub, int s) { return std::max(0, h_rng_hi(pl, tl, sub, s) - h_rng_lo(pl, tl, sub, s) + 1); }
// BEGIN
// sum of h_row_cells over the scores a .. b: the row's edges are piecewise linear in the score (each a min / max of three
// lines), so between two consecutive kinks the count is an arithmetic series
inline int64_t h_cells_sum(int pl, int tl, int sub, int a, int b) {
  if (b < a) return 0;
  const int64_t kinv = (int64_t)tl - pl, khi = kinv + sub, klo = kinv - sub;
  // scores at which two of the lines of an edge cross (the kink lies between the floor and the next integer)
  int64_t cand[16];
  int nc = 0;
  auto add = [&](int64_t x) { for (int64_t y : {x, x + 1}) if (y > a && y <= b) cand[nc++] = y; };
  add(tl); add(khi / 2 - (khi < 0 && (khi & 1) ? 1 : 0)); add(khi - tl);     // hi: s vs tl, s vs khi - s, tl vs khi - s
  add(pl); add((-klo) / 2 - (-klo < 0 && ((-klo) & 1) ? 1 : 0)); add(-klo - pl);  // lo: -s vs -pl, -s vs klo + s, -pl vs klo + s
  std::sort(cand, cand + nc);
  int64_t total = 0;
  int64_t u = a;
  auto cells = [&](int64_t s) { return (int64_t)h_rng_hi(pl, tl, sub, (int)s) - h_rng_lo(pl, tl, sub, (int)s) + 1; };
  auto seg = [&](int64_t x, int64_t y) {  // linear on [x, y]
    if (y < x) return;
    const int64_t cx = cells(x), cy = cells(y);
    if (cx <= 0 && cy <= 0) return;
    if (cx > 0 && cy > 0) { total += (cx + cy) * (y - x + 1) / 2; return; }
    if (y == x) { total += std::max<int64_t>(cx, 0); return; }
    // one end at or below zero: the slope is (cy - cx) / (y - x), an integer (each edge moves by whole diagonals per score)
    const int64_t slope = (cy - cx) / (y - x);
    if (cx > 0) {  // falls: positive up to x + (cx - 1) / -slope
      const int64_t last = x + (cx - 1) / (-slope);
      total += (cx + cells(last)) * (last - x + 1) / 2;
    } else {       // rises: positive from y - (cy - 1) / slope
      const int64_t first = y - (cy - 1) / slope;
      total += (cells(first) + cy) * (y - first + 1) / 2;
    }
  };
  for (int q = 0; q < nc; ++q) {
    if (cand[q] <= u) continue;
    seg(u, cand[q] - 1);
    u = cand[q];
  }
  seg(u, b);
  return total;
}
// END
int main() {
  srand(1);
  long bad = 0, n = 0;
  for (int it = 0; it < 400000; ++it) {
    int pl = rand() % 3000 + (rand() % 4 == 0 ? 0 : 1), tl = rand() % 3000 + 1;
    if (rand() % 3 == 0) { pl = rand() % 60 + 1; tl = rand() % 60 + 1; }
    int sub = rand() % 3 == 0 ? SUB_NONE : rand() % 4000;
    int a = rand() % 3500, b = a + rand() % 400 - 5;
    if (rand() % 5 == 0) { a = 0; b = rand() % 6000; }
    int64_t want = 0;
    for (int s = a; s <= b; ++s) want += h_row_cells(pl, tl, sub, s);
    int64_t got = h_cells_sum(pl, tl, sub, a, b);
    ++n;
    if (got != want) { if (bad < 10) printf("pl %d tl %d sub %d a %d b %d: got %lld want %lld\n", pl, tl, sub, a, b, (long long)got, (long long)want); ++bad; }
  }
  printf("%ld cases, %ld bad\n", n, bad);
  return bad != 0;
}
