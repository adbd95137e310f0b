// Plain C++ / OpenMP restatement of the reference LSTM-baseline train / eval step (TEST INFRASTRUCTURE ONLY: the second
// CPU-baseline variant of bench.py; checked against oracle/lstm_oracle.py in tests/test_cpu_ref.py).
// Graph: /root/reference/src/models/lstm_baseline.py:38-87 -- embedding_lookup (:39-42), BasicLSTMCell x L, static unroll
// over max_len, gate order i,j,f,o, forget_bias 1 added at run time (:44-55), xw_plus_b (:60-67), sequence_loss = mean
// sparse softmax cross entropy over all B*T tokens (:70-75), tf.gradients + clip_by_global_norm (:83-85, IndexedSlices
// norm for the embedding), exponential_decay + AdamOptimizer with TF's epsilon placement (:77-87).
// fp32 storage and arithmetic, rows time-major (row = t*B + b).  Never linked into the product.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <omp.h>

namespace {

struct Layer { int in; std::vector<float> K, b, mK, vK, mb, vb, dK, db; };      // K [(in+H)][4H], b [4H]

struct Model {
    int V1, T, E, H, L; float lr, clip, n_decay; int slices; long long step = 0;
    std::vector<float> emb, memb, vemb, demb, W, mW, vW, dW, d, md, vd, dd;
    std::vector<Layer> layers;
};

// C[M,N] (+)= A[M,K] * B[K,N]   (row-major; rows of C in parallel, the inner loop streams a row of B)
void gemm_nn(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc, bool accumulate) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < M; ++i) {
        float* c = C + (size_t)i * ldc;
        if (!accumulate) std::fill(c, c + N, 0.0f);
        for (int k = 0; k < K; ++k) {
            const float a = A[(size_t)i * lda + k];
            const float* b = B + (size_t)k * ldb;
#pragma omp simd
            for (int j = 0; j < N; ++j) c[j] += a * b[j];
        }
    }
}
// C[M,N] = A[M,K] * B[N,K]^T
void gemm_nt(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc) {
#pragma omp parallel for schedule(static) collapse(2)
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) {
            const float* a = A + (size_t)i * lda; const float* b = B + (size_t)j * ldb;
            float s = 0.0f;
#pragma omp simd reduction(+ : s)
            for (int k = 0; k < K; ++k) s += a[k] * b[k];
            C[(size_t)i * ldc + j] = s;
        }
}
// C[M,N] = A[K,M]^T * B[K,N]   (rows of C in parallel: each thread walks all K rows of B for its rows of C)
void gemm_tn(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < M; ++i) {
        float* c = C + (size_t)i * ldc;
        std::fill(c, c + N, 0.0f);
        for (int k = 0; k < K; ++k) {
            const float a = A[(size_t)k * lda + i];
            if (a == 0.0f) continue;
            const float* b = B + (size_t)k * ldb;
#pragma omp simd
            for (int j = 0; j < N; ++j) c[j] += a * b[j];
        }
    }
}
inline float sigm(float x) { return 1.0f / (1.0f + std::exp(-x)); }

struct Work {
    int B = 0; std::vector<int> X, Y;
    std::vector<std::vector<float>> xin, z, hs, cs;     // per layer: input [BT][in], gates [BT][4H], h/c [(T+1)B][H]
    std::vector<float> logits, dlogits, dh, dtop, dcur, dc, dhrec;
};

float forward(Model& m, Work& w, bool keep_logits) {
    const int B = w.B, T = m.T, H = m.H, G = 4 * m.H, n = B * T;
    w.xin.resize(m.L); w.z.resize(m.L); w.hs.resize(m.L); w.cs.resize(m.L);
    w.xin[0].resize((size_t)n * m.E);
#pragma omp parallel for
    for (int r = 0; r < n; ++r) std::memcpy(&w.xin[0][(size_t)r * m.E], &m.emb[(size_t)w.X[r] * m.E], sizeof(float) * m.E);
    for (int l = 0; l < m.L; ++l) {
        Layer& ly = m.layers[l];
        const int in = ly.in;
        w.z[l].resize((size_t)n * G); w.hs[l].assign((size_t)(T + 1) * B * H, 0.0f); w.cs[l].assign((size_t)(T + 1) * B * H, 0.0f);
        gemm_nn(n, G, in, w.xin[l].data(), in, ly.K.data(), G, w.z[l].data(), G, false);
        const float* Kh = ly.K.data() + (size_t)in * G;
        for (int t = 0; t < T; ++t) {
            float* zt = &w.z[l][(size_t)t * B * G];
            gemm_nn(B, G, H, &w.hs[l][(size_t)t * B * H], H, Kh, G, zt, G, true);
#pragma omp parallel for
            for (int b = 0; b < B; ++b)
                for (int u = 0; u < H; ++u) {
                    float* zr = zt + (size_t)b * G;
                    const float si = sigm(zr[u] + ly.b[u]), tj = std::tanh(zr[H + u] + ly.b[H + u]);
                    const float sf = sigm(zr[2 * H + u] + ly.b[2 * H + u] + 1.0f), so = sigm(zr[3 * H + u] + ly.b[3 * H + u]);
                    const float c = w.cs[l][((size_t)t * B + b) * H + u] * sf + si * tj;
                    w.cs[l][((size_t)(t + 1) * B + b) * H + u] = c;
                    w.hs[l][((size_t)(t + 1) * B + b) * H + u] = std::tanh(c) * so;
                    zr[u] = si; zr[H + u] = tj; zr[2 * H + u] = sf; zr[3 * H + u] = so;
                }
        }
        if (l + 1 < m.L) { w.xin[l + 1].assign(w.hs[l].begin() + (size_t)B * H, w.hs[l].end()); }
    }
    const float* out = &w.hs[m.L - 1][(size_t)B * H];
    w.logits.resize((size_t)n * m.V1);
    gemm_nn(n, m.V1, H, out, H, m.W.data(), m.V1, w.logits.data(), m.V1, false);
    double total = 0.0;
#pragma omp parallel for reduction(+ : total)
    for (int r = 0; r < n; ++r) {
        float* lg = &w.logits[(size_t)r * m.V1];
        float mx = -INFINITY;
        for (int v = 0; v < m.V1; ++v) { lg[v] += m.d[v]; mx = std::max(mx, lg[v]); }
        double s = 0.0;
        for (int v = 0; v < m.V1; ++v) s += std::exp((double)lg[v] - mx);
        const double lse = mx + std::log(s);
        total += lse - lg[w.Y[r]];
        if (keep_logits) {            // dlogits = (softmax - onehot) / (n + 1e-12), in place
            const float inv = (float)(1.0 / ((double)n + 1e-12));
            for (int v = 0; v < m.V1; ++v) lg[v] = (float)std::exp((double)lg[v] - lse) * inv;
            lg[w.Y[r]] -= inv;
        }
    }
    return (float)(total / ((double)n + 1e-12));
}

double backward(Model& m, Work& w) {          // returns the IndexedSlices squared norm of the embedding gradient
    const int B = w.B, T = m.T, H = m.H, G = 4 * m.H, n = B * T;
    const float* dl = w.logits.data();
    const float* out = &w.hs[m.L - 1][(size_t)B * H];
    gemm_tn(H, m.V1, n, out, H, dl, m.V1, m.dW.data(), m.V1);
    std::fill(m.dd.begin(), m.dd.end(), 0.0f);
    for (int r = 0; r < n; ++r) for (int v = 0; v < m.V1; ++v) m.dd[v] += dl[(size_t)r * m.V1 + v];
    w.dtop.resize((size_t)n * H);
    gemm_nt(n, H, m.V1, dl, m.V1, m.W.data(), m.V1, w.dtop.data(), H);
    double slices_sq = 0.0;
    for (int l = m.L - 1; l >= 0; --l) {
        Layer& ly = m.layers[l];
        const int in = ly.in;
        const float* Kh = ly.K.data() + (size_t)in * G;
        w.dc.assign((size_t)B * H, 0.0f); w.dhrec.assign((size_t)B * H, 0.0f);
        for (int t = T - 1; t >= 0; --t) {
            float* zt = &w.z[l][(size_t)t * B * G];
#pragma omp parallel for
            for (int b = 0; b < B; ++b)
                for (int u = 0; u < H; ++u) {
                    float* zr = zt + (size_t)b * G;
                    const float si = zr[u], tj = zr[H + u], sf = zr[2 * H + u], so = zr[3 * H + u];
                    const float ct = w.cs[l][((size_t)(t + 1) * B + b) * H + u], cp = w.cs[l][((size_t)t * B + b) * H + u];
                    const float dh = w.dtop[((size_t)t * B + b) * H + u] + w.dhrec[(size_t)b * H + u];
                    const float tc = std::tanh(ct);
                    const float dcv = w.dc[(size_t)b * H + u] + dh * so * (1.0f - tc * tc);
                    zr[u] = dcv * tj * si * (1.0f - si); zr[H + u] = dcv * si * (1.0f - tj * tj);
                    zr[2 * H + u] = dcv * cp * sf * (1.0f - sf); zr[3 * H + u] = dh * tc * so * (1.0f - so);
                    w.dc[(size_t)b * H + u] = dcv * sf;
                }
            gemm_nt(B, H, G, zt, G, Kh, G, w.dhrec.data(), H);
        }
        // dK = [xin, hprev]^T dz ; db = colsum dz ; dx = dz Kx^T
        gemm_tn(in, G, n, w.xin[l].data(), in, w.z[l].data(), G, ly.dK.data(), G);
        gemm_tn(H, G, n, w.hs[l].data(), H, w.z[l].data(), G, ly.dK.data() + (size_t)in * G, G);
        std::fill(ly.db.begin(), ly.db.end(), 0.0f);
        for (int r = 0; r < n; ++r) for (int c = 0; c < G; ++c) ly.db[c] += w.z[l][(size_t)r * G + c];
        w.dcur.resize((size_t)n * in);
        gemm_nt(n, in, G, w.z[l].data(), G, ly.K.data(), G, w.dcur.data(), in);
        if (l > 0) w.dtop = w.dcur;
    }
    std::fill(m.demb.begin(), m.demb.end(), 0.0f);
    for (int r = 0; r < n; ++r) {
        float* dst = &m.demb[(size_t)w.X[r] * m.E];
        const float* src = &w.dcur[(size_t)r * m.E];
        for (int e = 0; e < m.E; ++e) { dst[e] += src[e]; slices_sq += (double)src[e] * src[e]; }
    }
    return slices_sq;
}

void adam(std::vector<float>& p, std::vector<float>& mm, std::vector<float>& vv, const std::vector<float>& g, float scale, float alpha) {
    const long long n = (long long)p.size();
#pragma omp parallel for
    for (long long i = 0; i < n; ++i) {
        const float gc = g[i] * scale;
        mm[i] = 0.9f * mm[i] + 0.1f * gc;
        vv[i] = 0.999f * vv[i] + 0.001f * gc * gc;
        p[i] -= alpha * mm[i] / (std::sqrt(vv[i]) + 1e-8f);
    }
}
double sq(const std::vector<float>& g) { double s = 0; for (float x : g) s += (double)x * x; return s; }

void tokens(const Model& m, Work& w, const int32_t* sup, int ns, const int32_t* qry, int nq) {
    const int B = ns + nq, T = m.T;
    w.B = B; w.X.resize((size_t)B * T); w.Y.resize((size_t)B * T);
    for (int b = 0; b < B; ++b) {
        const int32_t* row = b < ns ? sup + (size_t)b * T : qry + (size_t)(b - ns) * T;
        for (int t = 0; t < T; ++t) {                    // base_model.py:63-86: target = the song, input = shifted right, start word first
            w.Y[(size_t)t * B + b] = row[t];
            w.X[(size_t)t * B + b] = t == 0 ? m.V1 - 1 : row[t - 1];
        }
    }
}

}  // namespace

extern "C" {

void* cpuref_create(int input_size, int max_len, int E, int H, int L, float lr, float clip, float n_decay, int slices_norm, int threads) {
    if (threads > 0) omp_set_num_threads(threads);
    Model* m = new Model();
    m->V1 = input_size + 1; m->T = max_len; m->E = E; m->H = H; m->L = L; m->lr = lr; m->clip = clip; m->n_decay = n_decay; m->slices = slices_norm;
    auto init = [](std::vector<float>& a, std::vector<float>& b, std::vector<float>& c, std::vector<float>& d, size_t n) { a.assign(n, 0.f); b.assign(n, 0.f); c.assign(n, 0.f); d.assign(n, 0.f); };
    init(m->emb, m->memb, m->vemb, m->demb, (size_t)m->V1 * E);
    init(m->W, m->mW, m->vW, m->dW, (size_t)H * m->V1);
    init(m->d, m->md, m->vd, m->dd, (size_t)m->V1);
    m->layers.resize(L);
    for (int l = 0; l < L; ++l) {
        Layer& ly = m->layers[l]; ly.in = l == 0 ? E : H;
        init(ly.K, ly.mK, ly.vK, ly.dK, (size_t)(ly.in + H) * 4 * H);
        init(ly.b, ly.mb, ly.vb, ly.db, (size_t)4 * H);
    }
    return m;
}
void cpuref_destroy(void* h) { delete (Model*)h; }

// name: "embedding", "kernel_<l>", "bias_<l>", "softmax_w", "softmax_b" (reference layouts); set = 1 writes, 0 reads
int cpuref_param(void* h, const char* name, float* buf, long long count, int set) {
    Model* m = (Model*)h; std::vector<float>* v = nullptr;
    if (!std::strcmp(name, "embedding")) v = &m->emb; else if (!std::strcmp(name, "softmax_w")) v = &m->W; else if (!std::strcmp(name, "softmax_b")) v = &m->d;
    else if (!std::strncmp(name, "kernel_", 7)) { int l = std::atoi(name + 7); if (l >= 0 && l < m->L) v = &m->layers[l].K; }
    else if (!std::strncmp(name, "bias_", 5)) { int l = std::atoi(name + 5); if (l >= 0 && l < m->L) v = &m->layers[l].b; }
    if (!v || (long long)v->size() != count) return -1;
    if (set) std::memcpy(v->data(), buf, sizeof(float) * count); else std::memcpy(buf, v->data(), sizeof(float) * count);
    return 0;
}

float cpuref_eval(void* h, const int32_t* query, int nq) {
    Model* m = (Model*)h; Work w; tokens(*m, w, query, nq, query, 0);
    return forward(*m, w, false);
}

float cpuref_train(void* h, const int32_t* support, int ns, const int32_t* query, int nq) {
    Model* m = (Model*)h; Work w; tokens(*m, w, support, ns, query, nq);
    const float loss = forward(*m, w, true);
    const double slices_sq = backward(*m, w);
    double s = (m->slices ? slices_sq : sq(m->demb)) + sq(m->dW) + sq(m->dd);
    for (auto& ly : m->layers) s += sq(ly.dK) + sq(ly.db);
    const double gnorm = std::sqrt(s);
    const float scale = (float)(m->clip / std::max(gnorm, (double)m->clip));
    const double t = (double)(m->step + 1);
    const double lr_s = (double)m->lr * std::pow(0.5, (double)m->step / (double)m->n_decay);
    const float alpha = (float)(lr_s * std::sqrt(1.0 - std::pow(0.999, t)) / (1.0 - std::pow(0.9, t)));
    adam(m->emb, m->memb, m->vemb, m->demb, scale, alpha);
    adam(m->W, m->mW, m->vW, m->dW, scale, alpha);
    adam(m->d, m->md, m->vd, m->dd, scale, alpha);
    for (auto& ly : m->layers) { adam(ly.K, ly.mK, ly.vK, ly.dK, scale, alpha); adam(ly.b, ly.mb, ly.vb, ly.db, scale, alpha); }
    m->step += 1;
    return loss;
}

}  // extern "C"
