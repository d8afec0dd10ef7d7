#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../emu/qs_emu.cpp"
int main(int argc, char** argv) {
    qs_config cfg; FILE* f = fopen(argv[1], "rb"); if (fread(&cfg, sizeof(cfg), 1, f) != 1) return 2; fclose(f);
    void* h = qse_create(&cfg);
    int n = cfg.n_envs, d = cfg.action_dim, o = cfg.obs_dim;
    std::vector<float> a((size_t)n * d), obs((size_t)n * o), rew(n), trace(70 * cfg.action_repeat);
    std::vector<unsigned char> dn(n), tr(n);
    qse_set_trace(h, n - 1, trace.data());
    std::vector<float> demo((size_t)37 * (d + 38));
    std::vector<int32_t> cnt(n);
    const bool is_demo = cfg.task >= QS_TASK_JUMPING_IN_PLACE_DEMO;
    if (is_demo) {
        for (size_t i = 0; i < demo.size(); i++) demo[i] = (float)(i % 13) / 13.0f - 0.5f;
        qse_set_demo(h, demo.data(), 37);
    }
    qse_reset(h, nullptr);
    if (is_demo) { for (int i = 0; i < n; i++) cnt[i] = (7 * i) % 37; qse_set_demo_counter(h, nullptr, cnt.data()); }
    unsigned s = 1; long dones = 0;
    for (int t = 0; t < atoi(argv[2]); t++) {
        for (int i = 0; i < n * d; i++) { s = s * 1664525u + 1013904223u; a[i] = ((t / 20) % 3 == 0) ? ((s >> 16) & 1 ? 1.2f : -1.2f) : ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
        qse_step(h, a.data(), obs.data(), rew.data(), dn.data(), tr.data());
        for (int i = 0; i < n; i++) dones += dn[i];
    }
    printf("ok dones=%ld obs0=%g\n", dones, obs[0]);
    qse_destroy(h);
    return 0;
}
