// Development aid: the embedding-table gradient (polygon_amd/csrc/pg_scatter.h, the code that ships) in isolation: checked
// against a host sum and timed, on tokens with a SMILES-like skew (a few symbols carry most rows).
//   hipcc -O3 --offload-arch=gfx950 -Ipolygon_amd/csrc -Iinclude tools/scatter_bench.hip -o tools/bin/scatter_bench
// (History: this tool is what showed that LDS float atomics retire about one lane per three clocks on gfx950 - 180 us for
// an LDS-table version of the kernel whatever its bank layout - and that a per-wave table with plain read-modify-write is
// bound by one LDS round trip per row, 56 us; see the header of pg_scatter.h.)
#include "pg_scatter.h"
#include <cstdlib>
#include <vector>
#include <cmath>
thread_local char g_pg_err[512] = {0};

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static float bf(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

int run_case(long N, int C, int Vp, size_t scratch_cap, int reps) {
  std::vector<int> tok(N);
  unsigned h = 12345u + (unsigned)N;
  for (long i = 0; i < N; ++i) {
    h = h * 1664525u + 1013904223u;
    const unsigned r = (h >> 8) % 100;
    tok[i] = r < 45 ? 3 : r < 65 ? 4 : r < 75 ? 5 : r < 82 ? 6 : 7 + (int)((h >> 16) % (Vp > 40 ? 22 : 3));
  }
  std::vector<unsigned short> g((size_t)N * C);
  for (size_t i = 0; i < g.size(); ++i) { h = h * 1664525u + 1013904223u; g[i] = 0x3c00 | ((h >> 9) & 0x3ff) | ((h >> 3) & 0x8000); }
  const size_t need = tok_scatter_layout(N, C, Vp).total;
  const size_t scratch_bytes = scratch_cap ? std::min(scratch_cap, need) : need;
  int* d_tok; bf16_t* d_g; char* d_scr; float *d_dT, *d_cs;
  CHECK(hipMalloc(&d_tok, N * 4));
  CHECK(hipMalloc(&d_g, g.size() * 2));
  CHECK(hipMalloc(&d_scr, scratch_bytes));
  CHECK(hipMalloc(&d_dT, (size_t)Vp * C * 4));
  CHECK(hipMalloc(&d_cs, (size_t)C * 4));
  CHECK(hipMemcpy(d_tok, tok.data(), N * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_g, g.data(), g.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemset(d_dT, 0xff, (size_t)Vp * C * 4));   // (the kernels must overwrite, not accumulate)
  CHECK(hipMemset(d_cs, 0, (size_t)C * 4));
  if (launch_tok_scatter_sum<bf16_t>(d_tok, d_g, C, N, C, Vp, d_dT, d_cs, d_scr, scratch_bytes, 0) != PG_OK) {
    printf("launch failed: %s\n", g_pg_err);
    return 1;
  }
  CHECK(hipDeviceSynchronize());
  std::vector<float> dT((size_t)Vp * C), cs(C);
  CHECK(hipMemcpy(dT.data(), d_dT, dT.size() * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(cs.data(), d_cs, cs.size() * 4, hipMemcpyDeviceToHost));
  std::vector<double> ref((size_t)Vp * C, 0.0), rcs(C, 0.0);
  for (long i = 0; i < N; ++i)
    for (int c = 0; c < C; ++c) { const double x = bf(g[(size_t)i * C + c]); ref[(size_t)tok[i] * C + c] += x; rcs[c] += x; }
  double err = 0, mag = 0, cerr = 0;
  for (size_t i = 0; i < ref.size(); ++i) { err = std::max(err, std::fabs(ref[i] - dT[i])); mag = std::max(mag, std::fabs(ref[i])); }
  for (int c = 0; c < C; ++c) cerr = std::max(cerr, std::fabs(rcs[c] - cs[c]));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch_tok_scatter_sum<bf16_t>(d_tok, d_g, C, N, C, Vp, d_dT, (float*)nullptr, d_scr, scratch_bytes, 0);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1000.0 / reps;
  printf("N %7ld C %5d Vp %3d scratch %6.1f MB (needs %6.1f): max err %.3g of %.3g, colsum err %.3g; %7.1f us per call (%.2f TB/s of G)\n",
         N, C, Vp, scratch_bytes / 1048576.0, need / 1048576.0, err, mag, cerr, us, (double)N * C * 2 / (us * 1e-6) / 1e12);
  fflush(stdout);
  hipFree(d_tok); hipFree(d_g); hipFree(d_scr); hipFree(d_dT); hipFree(d_cs);
  return (err <= 1e-3 * mag + 1e-3 && cerr <= 1e-3 * mag * 4 + 1e-2) ? 0 : 2;
}

int main() {
  int rc = 0;
  rc |= run_case(49000, 1536, 64, 0, 20);          // decoder of the bench configuration
  rc |= run_case(22000, 1536, 64, 0, 20);          // the shape of the header's history
  rc |= run_case(50000, 768, 64, 0, 20);           // encoder
  rc |= run_case(333, 96, 64, 0, 5);               // ragged: fewer rows than a chunk for most tokens, C not a multiple of 128
  rc |= run_case(50000, 768, 128, 0, 5);           // wide vocabulary
  rc |= run_case(300000, 96, 64, 0, 3);            // more rows than one sort launch takes (255 per thread): two passes
  rc |= run_case(22000, 1536, 64, (size_t)512 << 10, 5);   // scratch for ~2 k rows: eleven accumulating passes
  printf(rc ? "FAILED\n" : "ok\n");
  return rc;
}
