/* hipstub.c -- a DRY-RUN HIP runtime for profiling and testing the HOST side of libgtsam_amd.so in a container that has
 * no GPU (symbolic analysis, schedule construction, launch issue order, host cost per launch).
 *
 * TEST / PROFILING INFRASTRUCTURE ONLY.  It is never linked into, loaded by or shipped with the product: it only takes
 * effect when a developer preloads it explicitly (LD_PRELOAD=tools/hipstub/libhipstub.so, see tools/host_profile.py and
 * tests/test_host_analysis.py).  "Device" memory is host memory, copies are memcpy, kernels DO NOT RUN (hipLaunchKernel
 * only counts), so nothing numeric comes out of a process that runs under it -- it is not a CPU fallback.
 *
 * The entry points are the ones `nm -D --undefined-only gtsam_amd/lib/libgtsam_amd.so | grep hip` lists.
 * Counters: hipstub_launches(), hipstub_bytes_h2d(), hipstub_allocated() (plain C, read through ctypes).
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef struct { unsigned x, y, z; } dim3s;

static long long g_launches, g_h2d, g_alloc, g_streams, g_events;

/* every host-to-device copy is recorded as (bytes, FNV-1a hash of the bytes): two builds of the library that produce the same
 * multiset of records for the same problem have uploaded identical tables (tests/test_host_analysis.py) */
#define HIPSTUB_MAX_REC 8192
static struct { long long n; unsigned long long h; } g_rec[HIPSTUB_MAX_REC];
static int g_nrec;
static void record_h2d(const void* s, size_t n) {
  static int off = -1;   /* HIPSTUB_NO_HASH=1: timing runs (hashing ~100 MB of uploads costs tens of ms) */
  if (off < 0) off = getenv("HIPSTUB_NO_HASH") != NULL;
  if (off) return;
  const unsigned char* p = (const unsigned char*)s;
  unsigned long long h = 1469598103934665603ull;
  size_t i = 0;
  for (; i + 8 <= n; i += 8) { unsigned long long w; memcpy(&w, p + i, 8); h = (h ^ w) * 1099511628211ull; }
  for (; i < n; i++) h = (h ^ p[i]) * 1099511628211ull;
  int k = __atomic_fetch_add(&g_nrec, 1, __ATOMIC_RELAXED);
  if (k < HIPSTUB_MAX_REC) { g_rec[k].n = (long long)n; g_rec[k].h = h; }
}
int hipstub_h2d_count(void) { return g_nrec < HIPSTUB_MAX_REC ? g_nrec : HIPSTUB_MAX_REC; }
void hipstub_h2d_record(int i, long long* n, unsigned long long* h) { *n = g_rec[i].n; *h = g_rec[i].h; }

/* ---- operation trace (hipstub_trace_enable(1)): every launch, event record / wait and stream-ordered memory operation in issue
 * order, with the stream, the kernel (host stub address -> name from __hipRegisterFunction) and the first 8 argument words.
 * tools/race_check.py turns it into a happens-before graph and checks the multi-stream schedule of the Cholesky for races. */
enum { OP_LAUNCH = 1, OP_RECORD = 2, OP_WAIT = 3, OP_MEMOP = 4, OP_SYNC = 5 };
typedef struct { int type; unsigned grid; void* stream; const void* obj; unsigned long long args[8]; } hipstub_op;
#define HIPSTUB_MAX_OPS (1 << 20)
static hipstub_op* g_ops;
static int g_nops, g_trace;
#define HIPSTUB_MAX_FN 512
static struct { const void* host; const char* name; } g_fn[HIPSTUB_MAX_FN];
static int g_nfn;
void hipstub_trace_enable(int on) { if (on && !g_ops) g_ops = (hipstub_op*)calloc(HIPSTUB_MAX_OPS, sizeof(hipstub_op)); g_trace = on; if (on) g_nops = 0; }
int hipstub_trace_count(void) { return g_nops < HIPSTUB_MAX_OPS ? g_nops : HIPSTUB_MAX_OPS; }
const hipstub_op* hipstub_trace_ops(void) { return g_ops; }
const char* hipstub_kernel_name(const void* host) { for (int i = 0; i < g_nfn; i++) if (g_fn[i].host == host) return g_fn[i].name; return "?"; }
static hipstub_op* trace(int type, void* stream, const void* obj) {
  if (!g_trace) return NULL;
  int k = __atomic_fetch_add(&g_nops, 1, __ATOMIC_RELAXED);
  if (k >= HIPSTUB_MAX_OPS) return NULL;
  g_ops[k].type = type; g_ops[k].stream = stream; g_ops[k].obj = obj; g_ops[k].grid = 0;
  return &g_ops[k];
}

long long hipstub_launches(void) { return g_launches; }
long long hipstub_bytes_h2d(void) { return g_h2d; }
long long hipstub_allocated(void) { return g_alloc; }
void hipstub_reset(void) { g_launches = 0; g_h2d = 0; g_nrec = 0; }

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* ---- module registration (what hipcc's host stubs call at load time) ---- */
void** __hipRegisterFatBinary(const void* data) { (void)data; static void* h; return &h; }
void __hipUnregisterFatBinary(void** h) { (void)h; }
void __hipRegisterFunction(void** m, const void* host, char* dev, const char* name, unsigned tl, void* tid, void* bid,
                           void* bd, void* gd, int* ws) {
  (void)m; (void)dev; (void)tl; (void)tid; (void)bid; (void)bd; (void)gd; (void)ws;
  /* only the library's own kernels (namespace gt): torch's ROCm libraries register tens of thousands of theirs */
  if (name && strstr(name, "N2gt") && g_nfn < HIPSTUB_MAX_FN) { g_fn[g_nfn].host = host; g_fn[g_nfn].name = name; g_nfn++; }
}
void __hipRegisterVar(void** m, void* var, char* hv, char* dv, int ext, size_t size, int c, int g) {
  (void)m; (void)var; (void)hv; (void)dv; (void)ext; (void)size; (void)c; (void)g;
}

/* <<<>>> launches: push/pop of the call configuration, then hipLaunchKernel */
static __thread struct { dim3s g, b; size_t shmem; hipStream_t s; } g_cfg;
hipError_t __hipPushCallConfiguration(dim3s grid, dim3s block, size_t shmem, hipStream_t s) {
  g_cfg.g = grid; g_cfg.b = block; g_cfg.shmem = shmem; g_cfg.s = s; return 0;
}
hipError_t __hipPopCallConfiguration(dim3s* grid, dim3s* block, size_t* shmem, hipStream_t* s) {
  *grid = g_cfg.g; *block = g_cfg.b; *shmem = g_cfg.shmem; *s = g_cfg.s; return 0;
}
hipError_t hipLaunchKernel(const void* f, dim3s grid, dim3s block, void** args, size_t shmem, hipStream_t s) {
  (void)block; (void)shmem;
  __atomic_add_fetch(&g_launches, 1, __ATOMIC_RELAXED);
  hipstub_op* op = trace(OP_LAUNCH, s, f);
  if (op) {
    op->grid = grid.x;
    /* the kernels of the Cholesky schedule take (SMat S, int k, list, count | ...): keep the first four arguments (of S its base
     * pointer; other kernels have other, possibly shorter, argument lists and are only traced by name) */
    const char* name = hipstub_kernel_name(f);
    const int syrk = strstr(name, "k_syrk") != 0;
    if (strstr(name, "k_panel128") || syrk)
      for (int i = 0; i < 4; i++) { unsigned long long w = 0; memcpy(&w, args[i], (i == 1 || (i == 3 && syrk)) ? 4 : 8); op->args[i] = w; }
  }
  return 0;
}
hipError_t hipFuncSetAttribute(const void* f, int attr, int v) { (void)f; (void)attr; (void)v; return 0; }
hipError_t hipFuncGetAttributes(void* attr, const void* f) { (void)attr; (void)f; return 0; }   /* gtg_prewarm: nothing to load here */

/* ---- devices ---- */
hipError_t hipGetDeviceCount(int* n) { *n = 8; return 0; }   /* a node: one process per "device" in the multi-rank dry runs */
hipError_t hipSetDevice(int d) { (void)d; return 0; }
hipError_t hipGetLastError(void) { return 0; }
const char* hipGetErrorString(hipError_t e) { (void)e; return "hipstub: no error"; }
hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { if (lo) *lo = 0; if (hi) *hi = -1; return 0; }
/* hipDeviceProp_t (R0600) is large; the library reads multiProcessorCount only.  Fill the whole struct with a value
 * that is sensible for every int field it could look at (256 CUs). */
hipError_t hipGetDevicePropertiesR0600(void* prop, int dev) {
  (void)dev;
  int* p = (int*)prop;
  for (size_t i = 0; i < 1472 / sizeof(int); i++) p[i] = 256;
  return 0;
}

/* ---- memory ---- */
hipError_t hipMalloc(void** p, size_t n) {
  *p = calloc(n ? n : 1, 1);
  __atomic_add_fetch(&g_alloc, (long long)n, __ATOMIC_RELAXED);
  return *p ? 0 : 2;
}
hipError_t hipExtMallocWithFlags(void** p, size_t n, unsigned flags) { (void)flags; return hipMalloc(p, n); }
hipError_t hipFree(void* p) { free(p); return 0; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, int kind) {
  memcpy(d, s, n);
  if (kind == 1) { __atomic_add_fetch(&g_h2d, (long long)n, __ATOMIC_RELAXED); record_h2d(s, n); }
  return 0;
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int kind, hipStream_t st) { trace(OP_MEMOP, st, NULL); return hipMemcpy(d, s, n, kind); }
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, int kind, hipStream_t st) {
  (void)st; (void)kind;
  for (size_t r = 0; r < h; r++) memcpy((char*)d + r * dp, (const char*)s + r * sp, w);
  return 0;
}
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) { trace(OP_MEMOP, st, NULL); memset(d, v, n); return 0; }

/* ---- streams and events ---- */
static hipError_t new_stream(hipStream_t* s) { *s = malloc(8); g_streams++; return 0; }
hipError_t hipStreamCreate(hipStream_t* s) { return new_stream(s); }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned f) { (void)f; return new_stream(s); }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned f, int p) { (void)f; (void)p; return new_stream(s); }
hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, unsigned n, const unsigned* m) { (void)n; (void)m; return new_stream(s); }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return 0; }
hipError_t hipStreamSynchronize(hipStream_t s) { trace(OP_SYNC, s, NULL); return 0; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned f) { (void)f; trace(OP_WAIT, s, e); return 0; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = calloc(1, sizeof(double)); g_events++; return 0; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned f) { (void)f; return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return 0; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { trace(OP_RECORD, s, e); if (e) *(double*)e = now_ms(); return 0; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(*(double*)b - *(double*)a); return 0; }
hipError_t hipEventSynchronize(hipEvent_t e) { (void)e; return 0; }
hipError_t hipDeviceSynchronize(void) { return 0; }
hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
int hipstub_kernel_count(void) { return g_nfn; }
const char* hipstub_kernel_at(int i, const void** host) { *host = g_fn[i].host; return g_fn[i].name; }
