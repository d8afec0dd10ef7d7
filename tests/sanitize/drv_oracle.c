#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "qso.h"
int main(int argc, char** argv) {
    qso_config cfg; FILE* f = fopen(argv[1], "rb"); if (fread(&cfg, sizeof(cfg), 1, f) != 1) return 2; fclose(f);
    qso_handle* h; if (qso_create(&cfg, &h)) { printf("create failed\n"); return 1; }
    int n = cfg.n_envs, d = cfg.action_dim, o = cfg.obs_dim;
    float* a = calloc((size_t)n * d, 4); float* obs = calloc((size_t)n * o, 4); float* rew = calloc(n, 4);
    unsigned char* dn = calloc(n, 1); unsigned char* tr = calloc(n, 1);
    double* trace = calloc(70 * cfg.action_repeat, sizeof(double));
    qso_set_trace(h, n - 1, trace);
    float* demo = NULL; int* cnt = calloc(n, sizeof(int));
    if (cfg.task >= QSO_TASK_JUMPING_IN_PLACE_DEMO) {   /* a synthetic 37-row demonstration; two installs exercise the replacement */
        demo = calloc((size_t)37 * (d + 38), 4);
        for (int i = 0; i < 37 * (d + 38); i++) demo[i] = (float)(i % 13) / 13.0f - 0.5f;
        qso_set_demo(h, demo, 20); qso_set_demo(h, demo, 37);
    }
    qso_reset(h, NULL); qso_get_obs(h, obs);
    if (demo) { for (int i = 0; i < n; i++) cnt[i] = (7 * i) % 37; qso_set_demo_counter(h, NULL, cnt); }
    unsigned s = 1; long dones = 0;
    for (int t = 0; t < atoi(argv[2]); t++) {
        for (int i = 0; i < n * d; i++) { s = s * 1664525u + 1013904223u; a[i] = ((t / 20) % 3 == 0) ? ((s >> 16) & 1 ? 1.2f : -1.2f) : ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
        qso_step(h, a, obs, rew, dn, tr);
        for (int i = 0; i < n; i++) dones += dn[i];
    }
    double info[64 * 48];
    for (int w = 0; w <= 10; w++) if (w != 9) qso_get_info(h, w, info);
    printf("ok dones=%ld obs0=%g\n", dones, obs[0]);
    qso_destroy(h); free(a); free(obs); free(rew); free(dn); free(tr); free(trace); free(demo); free(cnt);
    return 0;
}
