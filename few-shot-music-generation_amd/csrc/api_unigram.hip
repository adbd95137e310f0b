// Unigram baseline on the device (reference src/models/unigram_model.py:26-39).
// Host-side C++ only (part of the C-ABI of libfsmg, include/fsmg.h); every kernel lives in gemm.hip / lstm_*.hip / elementwise.hip.
#include "fsmg_model.h"

using namespace fsmg;
using namespace fsmg_host;


// =========================================================================== C ABI
extern "C" {

struct fsmg_unigram {
    int V = 0, device = 0;
    hipStream_t stream = nullptr;
    unsigned* counts = nullptr;
    int* words = nullptr; int64_t words_cap = 0;
    float* out = nullptr;           // [0] nll, [1] sum of counts; then an int: argmax; then an int: error flag
    std::string err;
};
namespace {
int ufail(fsmg_unigram* u, int code, const std::string& msg) { if (u) u->err = msg; else g_create_error = msg; return code; }
#define UCK(u, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return ufail(u, FSMG_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
int unigram_stage(fsmg_unigram* u, const int32_t* words, int64_t n, int on_device, const int** dev) {
    if (on_device) { *dev = words; return FSMG_OK; }
    if (u->words_cap < n) {
        UCK(u, hipStreamSynchronize(u->stream));
        if (u->words) hipFree(u->words);
        u->words = nullptr; u->words_cap = 0;
        const int64_t cap = std::max<int64_t>(n, 1 << 16);
        if (hipMalloc((void**)&u->words, sizeof(int) * (size_t)cap) != hipSuccess) return ufail(u, FSMG_ERR_NOMEM, "hipMalloc(words) failed");
        u->words_cap = cap;
    }
    UCK(u, hipMemcpyAsync(u->words, words, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, u->stream));
    *dev = u->words;
    return FSMG_OK;
}
int unigram_read(fsmg_unigram* u, float* nll) {
    float host[4] = {0.f, 0.f, 0.f, 0.f};
    UCK(u, hipMemcpyAsync(host, u->out, sizeof(host), hipMemcpyDeviceToHost, u->stream));
    UCK(u, hipStreamSynchronize(u->stream));
    int flag; std::memcpy(&flag, &host[3], 4);
    if (flag != 0) {
        UCK(u, hipMemsetAsync(u->out + 3, 0, 4, u->stream));
        return ufail(u, FSMG_ERR_TOKEN_RANGE, "word id outside [0, input_size)");
    }
    if (nll) *nll = host[0];
    return FSMG_OK;
}
}  // namespace

int fsmg_unigram_create(int32_t input_size, int32_t device, fsmg_unigram_handle* out) {
    if (!out || input_size <= 0) return ufail(nullptr, FSMG_ERR_INVALID, "bad input_size / out");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ufail(nullptr, FSMG_ERR_NO_DEVICE, "no HIP device visible: libfsmg has no CPU fallback");
    if (device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) return ufail(nullptr, FSMG_ERR_NO_DEVICE, "device ordinal out of range");
    fsmg_unigram* u = new (std::nothrow) fsmg_unigram();
    if (!u) return ufail(nullptr, FSMG_ERR_NOMEM, "host allocation failed");
    u->V = input_size; u->device = device;
    if (hipStreamCreateWithFlags(&u->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void**)&u->counts, sizeof(unsigned) * (size_t)input_size) != hipSuccess ||
        hipMalloc((void**)&u->out, 256) != hipSuccess) { fsmg_unigram_destroy(u); return ufail(nullptr, FSMG_ERR_NOMEM, "device allocation failed"); }
    hipMemsetAsync(u->out, 0, 256, u->stream);
    if (launch_fill32(u->stream, u->counts, 1u, input_size) != hipSuccess || hipStreamSynchronize(u->stream) != hipSuccess) {     // alpha = 1
        fsmg_unigram_destroy(u); return ufail(nullptr, FSMG_ERR_HIP, "count initialisation failed");
    }
    *out = u;
    return FSMG_OK;
}
int fsmg_unigram_destroy(fsmg_unigram_handle u) {
    if (!u) return FSMG_OK;
    hipSetDevice(u->device);
    if (u->stream) hipStreamSynchronize(u->stream);
    if (u->counts) hipFree(u->counts);
    if (u->words) hipFree(u->words);
    if (u->out) hipFree(u->out);
    if (u->stream) hipStreamDestroy(u->stream);
    delete u;
    return FSMG_OK;
}
const char* fsmg_unigram_last_error(fsmg_unigram_handle u) { return u ? u->err.c_str() : g_create_error.c_str(); }
int fsmg_unigram_nll(fsmg_unigram_handle u, const int32_t* words, int64_t n, int32_t on_device, float* nll) {
    if (!u || !words || n <= 0 || !nll) return FSMG_ERR_INVALID;
    hipSetDevice(u->device);
    const int* dev = nullptr;
    int rc = unigram_stage(u, words, n, on_device, &dev);
    if (rc != FSMG_OK) return rc;
    UCK(u, launch_unigram_nll(u->stream, dev, n, u->counts, u->V, u->out, (int*)(u->out + 3)));
    return unigram_read(u, nll);
}
int fsmg_unigram_train(fsmg_unigram_handle u, const int32_t* words, int64_t n, int32_t on_device, float* loss) {
    if (!u || !words || n <= 0) return FSMG_ERR_INVALID;
    hipSetDevice(u->device);
    const int* dev = nullptr;
    int rc = unigram_stage(u, words, n, on_device, &dev);
    if (rc != FSMG_OK) return rc;
    // the loss with the counts BEFORE the update, like LSTMBaseline.train's pre-update loss; a batch with an id out of range
    // is rejected as a whole (the NLL kernel has seen every word before the update runs)
    UCK(u, launch_unigram_nll(u->stream, dev, n, u->counts, u->V, u->out, (int*)(u->out + 3)));
    float l = 0.f;
    rc = unigram_read(u, &l);
    if (rc != FSMG_OK) return rc;
    UCK(u, launch_unigram_update(u->stream, dev, n, u->counts, u->V, (int*)(u->out + 3)));
    if (!on_device) UCK(u, hipStreamSynchronize(u->stream));      // the staging buffer may be reused by the next call
    if (loss) *loss = l;
    return FSMG_OK;
}
int fsmg_unigram_get_counts(fsmg_unigram_handle u, float* host, int64_t count) {
    if (!u || !host || count != u->V) return FSMG_ERR_INVALID;
    hipSetDevice(u->device);
    std::vector<unsigned> tmp((size_t)count);
    UCK(u, hipStreamSynchronize(u->stream));
    UCK(u, hipMemcpy(tmp.data(), u->counts, sizeof(unsigned) * (size_t)count, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < count; ++i) host[i] = (float)tmp[(size_t)i];
    return FSMG_OK;
}
int fsmg_unigram_set_counts(fsmg_unigram_handle u, const float* host, int64_t count) {
    if (!u || !host || count != u->V) return FSMG_ERR_INVALID;
    hipSetDevice(u->device);
    std::vector<unsigned> tmp((size_t)count);
    for (int64_t i = 0; i < count; ++i) {
        if (!(host[i] >= 0.f) || host[i] > 4.0e9f) return ufail(u, FSMG_ERR_INVALID, "counts must be finite and >= 0");
        // the counts live on the device as unsigned integers: a checkpoint whose counts are not whole numbers is refused, not
        // rounded behind the caller's back
        if (host[i] != std::floor(host[i])) return ufail(u, FSMG_ERR_INVALID, "counts must be whole numbers (element " + std::to_string(i) + " is " + std::to_string(host[i]) + ")");
        tmp[(size_t)i] = (unsigned)std::llround((double)host[i]);
    }
    UCK(u, hipStreamSynchronize(u->stream));
    UCK(u, hipMemcpy(u->counts, tmp.data(), sizeof(unsigned) * (size_t)count, hipMemcpyHostToDevice));
    return FSMG_OK;
}
int fsmg_unigram_argmax(fsmg_unigram_handle u, int32_t* word) {
    if (!u || !word) return FSMG_ERR_INVALID;
    hipSetDevice(u->device);
    UCK(u, launch_unigram_argmax(u->stream, u->counts, u->V, (int*)(u->out + 2)));
    int w = 0;
    UCK(u, hipMemcpyAsync(&w, u->out + 2, sizeof(int), hipMemcpyDeviceToHost, u->stream));
    UCK(u, hipStreamSynchronize(u->stream));
    *word = w;
    return FSMG_OK;
}

}  // extern "C"
