// Error channel and version for libksr.
#include "ksr_common.h"

#include <cstdlib>
#include <stdarg.h>
#include <map>
#include <mutex>

static thread_local char g_err[512] = "";

void ksr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* ksr_last_error(void) { return g_err; }

extern "C" int ksr_version(void) { return 100; }

// Precision mode: 0 = exact fp32, 1 = bf16 MFMA operands / fp32 accumulate and storage, 2 = as 1 with the
// activation and gradient tensors between kernels stored as bf16.
// Per calling thread (a new thread starts in mode 0), read at launch time: the caller sets it before a forward /
// backward sequence, and two models of different precision driven from two threads do not disturb each other.
static thread_local int g_prec = 0;
int ksr_prec() { return g_prec; }

// fp32 products of the forward kernels on the bf16 matrix cores (three-term split of both operands, as accurate against a
// float64 evaluation as the fp32 matrix instruction): on unless the environment holds KSR_F32_SPLIT=0; a thread can override
// the process default for itself.  Read by the fp32 entry points that have such an instance.
static thread_local int g_f32_split = -1;      // -1: the process default
static bool f32_split_default() {
    static const bool on = [] { const char* e = getenv("KSR_F32_SPLIT"); return !(e && e[0] == '0'); }();
    return on;
}
bool ksr_f32_split() { return g_f32_split < 0 ? f32_split_default() : g_f32_split != 0; }
extern "C" int ksr_set_f32_split(int on) { g_f32_split = on < 0 ? -1 : (on != 0); return KSR_OK; }
extern "C" int ksr_get_f32_split(void) { return ksr_f32_split() ? 1 : 0; }

extern "C" int ksr_set_precision(int prec) {
    KSR_CHECK_ARG(prec >= 0 && prec <= 2, "set_precision: 0 (fp32), 1 (bf16 MFMA operands) or 2 (bf16 operands and activation storage)");
    g_prec = prec;
    return KSR_OK;
}

extern "C" int ksr_get_precision(void) { return g_prec; }

// Per calling thread, like the precision mode: NULL (the default) = the by-value seeds are used as they are.
static thread_local const unsigned long long* g_seed_dev = nullptr;
const unsigned long long* ksr_seed_dev() { return g_seed_dev; }

extern "C" int ksr_set_dropout_seed_source(const uint64_t* device_word) {
    g_seed_dev = reinterpret_cast<const unsigned long long*>(device_word);
    return KSR_OK;
}

long ksr_wg_slots(const void* kernel, int block) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, long> cache;      // one process drives identical GPUs
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(kernel, block);
    const auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int per_cu = 0, dev = 0, cus = 0;
    long slots = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0) == hipSuccess && hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
        slots = (long)per_cu * cus;
    else
        (void)hipGetLastError();       // an unknown occupancy is not an error: the caller falls back to its cap
    cache[key] = slots;
    return slots;
}
