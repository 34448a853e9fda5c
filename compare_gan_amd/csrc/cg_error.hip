// Thread-local error string + ABI version.
#include <stdarg.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include "cg_common.h"

static thread_local char g_err[512] = {0};

void cg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int cg_abi_version(void) { return 7; }   // round 6: + cg_sum4, cg_gconv_ld (round 5: cgDeferCtx replaces cg_reduce_defer_*)
extern "C" const char* cg_last_error(void) { return g_err; }

// ---- optional per-kernel-family timing with HIP events (bench.py's roofline leg) -------------
namespace {
struct ProfSlot {
  std::vector<hipEvent_t> ev;   // start/stop pairs
  std::vector<double> flops, bytes;
  std::vector<std::string> tags;   // geometry of the launch (CGAMD_PROF_LOG)
  double total_ms = 0, total_flops = 0, total_bytes = 0;
  int64_t launches = 0;
};
bool g_prof_on = false;
ProfSlot g_prof[CG_PROF_COUNT];
const char* const g_prof_names[CG_PROF_COUNT] = {
    "halo_conv_kernel<*>",
    "hconv_kernel<128, *>",          "hconv_kernel<64, *>",
    "fast_conv_kernel<128, 128, *>", "fast_conv_kernel<64, 128, *>", "fast_conv_kernel<128, 64, *>",
    "fast_conv_kernel<128, 32, *>",  "stem_fwd_kernel<*>",           "gconv_kernel<...>",
    "hwgrad_kernel<*>",              "halo_wgrad_kernel<*>",
    "fast_wgrad_kernel<128, *>",     "fast_wgrad_kernel<64, *>",     "stem_wgrad_kernel<*>",
    "gwgrad_kernel<...>",            "sconv_kernel<*>",              "swgrad_kernel<*>",
    "fast_conv_kernel<128, 192, *>"};
}  // namespace

bool cg_prof_enabled() { return g_prof_on; }

// geometry tag of the NEXT cg_prof_begin on this thread (written to CGAMD_PROF_LOG at collect time)
static thread_local char g_prof_tag[128] = {0};
void cg_prof_tag_geom(const cgConvGeom* g) {
  snprintf(g_prof_tag, sizeof(g_prof_tag), "N%d %dx%dx%d->%dx%dx%d k%dx%d S%d U%d", g->N, g->Hin,
           g->Win, g->Ci, g->Ho, g->Wo, g->Co, g->kh, g->kw, g->S, g->U);
}

void cg_prof_begin(int family, double flops, double bytes, hipStream_t st) {
  if (!g_prof_on || family < 0 || family >= CG_PROF_COUNT) return;
  ProfSlot& p = g_prof[family];
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  p.ev.push_back(a);
  p.ev.push_back(b);
  p.flops.push_back(flops);
  p.bytes.push_back(bytes);
  p.tags.push_back(g_prof_tag);
  g_prof_tag[0] = 0;
  hipEventRecord(a, st);
}

void cg_prof_end(int family, hipStream_t st) {
  if (!g_prof_on || family < 0 || family >= CG_PROF_COUNT) return;
  ProfSlot& p = g_prof[family];
  if (p.ev.size() >= 2) hipEventRecord(p.ev.back(), st);
}

// CRC32C (Castagnoli, reflected 0x82F63B78), slice-by-8 on the host: the checksum of TFRecord
// payloads (datasets.py:430-532 reads TFDS shards) and of tensor-bundle blocks / tensors
// (compare_gan_amd/tf_checkpoint.py).  Plain host code: no device, no stream.
extern "C" uint32_t cg_host_crc32c(const void* data, size_t n, uint32_t seed) {
  static uint32_t T[8][256];
  static bool ready = [] {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      T[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int t = 1; t < 8; ++t) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 0xFF];
    return true;
  }();
  (void)ready;
  const unsigned char* p = (const unsigned char*)data;
  uint32_t c = ~seed;
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    w ^= c;
    c = T[7][w & 0xFF] ^ T[6][(w >> 8) & 0xFF] ^ T[5][(w >> 16) & 0xFF] ^ T[4][(w >> 24) & 0xFF] ^
        T[3][(w >> 32) & 0xFF] ^ T[2][(w >> 40) & 0xFF] ^ T[1][(w >> 48) & 0xFF] ^ T[0][w >> 56];
    p += 8;
    n -= 8;
  }
  while (n--) c = T[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return ~c;
}

extern "C" int cg_prof_family_count(void) { return CG_PROF_COUNT; }
extern "C" const char* cg_prof_family_name(int family) {
  return (family >= 0 && family < CG_PROF_COUNT) ? g_prof_names[family] : "";
}

extern "C" int cg_prof_enable(int on) {
  g_prof_on = on != 0;
  return CG_OK;
}

extern "C" int cg_prof_collect(int family, double* total_ms, int64_t* launches, double* flops,
                               double* bytes) {
  if (family < 0 || family >= CG_PROF_COUNT) CG_FAIL(CG_ERR_BAD_ARG, "cg_prof_collect: family");
  ProfSlot& p = g_prof[family];
  const char* log_path = getenv("CGAMD_PROF_LOG");   // per-launch lines: family;geometry;us;GFLOP
  FILE* log = (log_path && *log_path && !p.ev.empty()) ? fopen(log_path, "a") : nullptr;
  for (size_t i = 0; i + 1 < p.ev.size(); i += 2) {
    hipEventSynchronize(p.ev[i + 1]);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.ev[i], p.ev[i + 1]) == hipSuccess) {
      if (log)
        fprintf(log, "%s;%s;%.2f;%.3f\n", g_prof_names[family], p.tags[i / 2].c_str(), ms * 1e3,
                p.flops[i / 2] * 1e-9);
      p.total_ms += ms;
      p.total_flops += p.flops[i / 2];
      p.total_bytes += p.bytes[i / 2];
      p.launches += 1;
    }
    hipEventDestroy(p.ev[i]);
    hipEventDestroy(p.ev[i + 1]);
  }
  if (log) fclose(log);
  p.ev.clear();
  p.flops.clear();
  p.bytes.clear();
  p.tags.clear();
  if (total_ms) *total_ms = p.total_ms;
  if (launches) *launches = p.launches;
  if (flops) *flops = p.total_flops;
  if (bytes) *bytes = p.total_bytes;
  return CG_OK;
}

extern "C" int cg_prof_reset(void) {
  for (int f = 0; f < CG_PROF_COUNT; ++f) {
    cg_prof_collect(f, nullptr, nullptr, nullptr, nullptr);
    g_prof[f].total_ms = g_prof[f].total_flops = g_prof[f].total_bytes = 0;
    g_prof[f].launches = 0;
  }
  return CG_OK;
}
