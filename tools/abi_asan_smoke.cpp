// Drives the C-ABI from plain C++ (no Python, no torch) so that the host shim can run under AddressSanitizer:
//   hipcc -fsanitize=address -shared-libsan -g -Iinclude tools/abi_asan_smoke.cpp -Lfew-shot-music-generation_amd/lib -lfsmg_asan -o tools/abi_asan_smoke.bin
//   ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 LD_LIBRARY_PATH=few-shot-music-generation_amd/lib ./tools/abi_asan_smoke.bin
// (PyTorch-ROCm exits silently at import under the ASan runtime, so the pytest suite runs under UBSan only.)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <unistd.h>
#include "fsmg.h"
#define CK(x) do { int rc_ = (x); if (rc_ != 0) { printf("%s -> %d: %s\n", #x, rc_, fsmg_last_error(h)); return 1; } } while (0)

int main() {
    fsmg_handle h = nullptr;
    for (int hidden : {40, 512}) {
        fsmg_config c; std::memset(&c, 0, sizeof(c));
        c.input_size = 97; c.max_len = 9; c.embedding_size = 20; c.hidden_size = hidden; c.n_layers = hidden == 40 ? 2 : 1;
        c.lr = 5e-3f; c.max_grad_norm = 5.f; c.n_decay = 100.f; c.use_graph = 1; c.config_version = FSMG_CONFIG_VERSION;
        if (fsmg_create(&c, &h) != 0) { printf("create: %s\n", fsmg_last_error(nullptr)); return 1; }
        CK(fsmg_init_params(h, 7));
        const int N = 3, K = 2, Q = 2, T = c.max_len;
        std::vector<int32_t> sup(N * K * T), qry(N * Q * T), table(50 * T), si(N * K), qi(N * Q);
        srand(1);
        for (auto& v : sup) v = rand() % 97; for (auto& v : qry) v = rand() % 97; for (auto& v : table) v = rand() % 97;
        for (auto& v : si) v = rand() % 50; for (auto& v : qi) v = rand() % 50;
        float loss = 0, nll = 0;
        for (int s = 0; s < 3; ++s) CK(fsmg_train_step(h, sup.data(), qry.data(), N, K, Q, 0, &loss));
        CK(fsmg_eval_step(h, qry.data(), N, Q, 0, &nll));
        std::vector<int32_t> many(4 * N * Q * T); for (auto& v : many) v = rand() % 97;
        std::vector<float> nlls(4);
        CK(fsmg_eval_batch(h, many.data(), 4, N, Q, 0, nlls.data()));
        CK(fsmg_maml_step(h, sup.data(), qry.data(), N, K, Q, 2, 0.1f, 0, &loss));
        CK(fsmg_maml_eval(h, sup.data(), qry.data(), N, K, Q, 1, 0.1f, 0, &nll));
        CK(fsmg_upload_table(h, 0, table.data(), 50));
        CK(fsmg_train_step_indexed(h, 0, si.data(), qi.data(), N, K, Q, &loss));
        CK(fsmg_train_step_indexed(h, 0, si.data(), qi.data(), N, K, Q, nullptr));
        std::vector<int32_t> toks(12); CK(fsmg_sample(h, 12, toks.data()));
        char name[64]; int64_t rows, cols;
        for (int i = 0; i < fsmg_num_params(h); ++i) {
            CK(fsmg_param_info(h, i, name, sizeof(name), &rows, &cols));
            std::vector<float> p(rows * cols), m(rows * cols), v(rows * cols);
            CK(fsmg_get_param(h, name, p.data(), rows * cols)); CK(fsmg_get_grad(h, name, m.data(), rows * cols));
            CK(fsmg_get_opt_state(h, name, m.data(), v.data(), rows * cols)); CK(fsmg_set_opt_state(h, name, m.data(), v.data(), rows * cols));
            CK(fsmg_set_param(h, name, p.data(), rows * cols));
        }
        float last[3]; CK(fsmg_read_losses(h, last, 3));
        fsmg_stats st; CK(fsmg_get_stats(h, &st));
        int64_t step; CK(fsmg_get_step(h, &step));
        sup[0] = 97;                                                   // out-of-range token: reported, update skipped
        const int rc = fsmg_train_step(h, sup.data(), qry.data(), N, K, Q, 0, &loss);
        printf("hidden %d: loss %.4f nll %.4f step %lld xcd launches %lld timeouts %lld, bad token -> rc %d (%s)\n", hidden, loss, nll,
               (long long)step, (long long)st.xcd_launches, (long long)st.timeouts, rc, fsmg_last_error(h));
        if (rc != FSMG_ERR_TOKEN_RANGE) return 1;
        CK(fsmg_destroy(h)); h = nullptr;
    }
    printf("ABI_ASAN_SMOKE_OK\n");
    fflush(stdout);
    _exit(0);          // skip the HIP runtime's exit-time finalizers: ROCm 7.2's ASan device allocator CHECK-fails in them (not in this library)
}
