// Branch coverage of the emulated kernels (test infrastructure): linked into build_cov/emu_render, whose kernel
// object is compiled with -fsanitize-coverage=trace-pc-guard,pc-table.  At exit the program writes to the file
// "$EMU_COVERAGE_FILE.<pid>" one line per instrumented edge: "<pc in hex> <1 if it ran, else 0>".
// tools/emu_coverage_report.py merges such files and names the source lines that never ran.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <vector>

namespace {
uint32_t* g_guard_begin = nullptr;
uint32_t* g_guard_end = nullptr;
const uintptr_t* g_pcs_begin = nullptr;
const uintptr_t* g_pcs_end = nullptr;
std::vector<unsigned char>* g_hit = nullptr;

void dump() {
  const char* path = getenv("EMU_COVERAGE_FILE");
  if (!path || !g_hit || !g_pcs_begin) return;
  char name[4096];
  snprintf(name, sizeof name, "%s.%ld", path, (long)getpid());  // one file per process: runs may be concurrent
  FILE* f = fopen(name, "w");
  if (!f) return;
  // the pc table holds (pc, flags) pairs, one per guard; the program is linked -no-pie, so a pc is a file address
  size_t n = (size_t)(g_guard_end - g_guard_begin);
  const size_t n_pcs = (size_t)(g_pcs_end - g_pcs_begin) / 2;
  if (n_pcs < n) n = n_pcs;
  for (size_t i = 0; i < n; ++i) fprintf(f, "%lx %d\n", (unsigned long)g_pcs_begin[2 * i], (int)(*g_hit)[i]);
  fclose(f);
}
}  // namespace

extern "C" void __sanitizer_cov_trace_pc_guard_init(uint32_t* start, uint32_t* stop) {
  if (start == stop || *start) return;
  g_guard_begin = start;
  g_guard_end = stop;
  uint32_t n = 0;
  for (uint32_t* g = start; g < stop; ++g) *g = ++n;  // guard value = edge index + 1
  g_hit = new std::vector<unsigned char>(n, 0);
  atexit(dump);
}

extern "C" void __sanitizer_cov_pcs_init(const uintptr_t* begin, const uintptr_t* end) {
  g_pcs_begin = begin;
  g_pcs_end = end;
}

extern "C" void __sanitizer_cov_trace_pc_guard(uint32_t* guard) {
  const uint32_t g = *guard;
  if (!g) return;
  (*g_hit)[g - 1] = 1;
  *guard = 0;  // later visits of this edge return at once
}
