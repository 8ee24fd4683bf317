/* enhance_wav.c -- the whole predict path of the reference (src/predict.py -> SGMSEModule.predict_step -> ScoreModel.sample) for one
 * file through the C ABI of libuse_hip.so alone: no Python, no torch.  What a non-Python host of the library looks like.
 *
 *   enhance_wav <weights.usehip> <noisy.wav> <enhanced.wav> [N=30] [seed=0] [precision: bf16|fp16|fp32 = what the file was packed for]
 *
 * weights.usehip: `python -m universal_speech_enhancement_amd.pack_checkpoint ckpt=last.ckpt out=weights.usehip precision=bf16`
 * Steps (reference file:line): loader (loadwav_dataset.py:90-120) -> STFT + compression + padding (model_wrapper.py:275-278) ->
 * 30-step PC sampler, reverse diffusion + Langevin x1, snr 0.5 (SGMSE_Large.yaml, sampling/__init__.py:59-71) -> decompression +
 * iSTFT (:320) -> sf.write (SGMSE_module.py:80).
 * build: gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/enhance_wav.c -o examples/enhance_wav \
 *            -Luniversal_speech_enhancement_amd -luse_hip -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,'$ORIGIN/../universal_speech_enhancement_amd' */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "use_hip.h"

#define CHECK(call)                                                                          \
    do {                                                                                     \
        int rc_ = (call);                                                                    \
        if (rc_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, use_last_error()); return 1; } \
    } while (0)
#define HIPCHECK(call)                                                                       \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_)); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s weights.usehip noisy.wav enhanced.wav [N=30] [seed=0] [bf16|fp16|fp32]\n", argv[0]); return 2; }
    const int N = argc > 4 ? atoi(argv[4]) : 30;
    const unsigned long long seed = argc > 5 ? strtoull(argv[5], NULL, 10) : 0ull;
    const char* prec = argc > 6 ? argv[6] : "bf16";
    const int n_fft = 1022, hop = 160, F = n_fft / 2 + 1;
    const float factor = 0.15f, expo = 0.5f;

    float* wav = NULL; int64_t L = 0; int sr = 0;
    CHECK(use_load_utterance(argv[2], 24000, 1, &wav, &L, &sr));           /* first channel, 24 kHz FFT resampling, peak 0.8 */
    const int T = 1 + (int)(L / hop), Tpad = (T + 63) / 64 * 64;

    use_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.nf = 128; cfg.n_levels = 7; cfg.num_res_blocks = 2; cfg.n_freq = F;
    { const int cm[7] = {1, 1, 2, 2, 2, 2, 2}; memcpy(cfg.ch_mult, cm, sizeof cm); }
    cfg.precision = !strcmp(prec, "fp32") ? USE_PREC_FP32 : !strcmp(prec, "fp16") ? USE_PREC_FP16 : USE_PREC_BF16;
    cfg.theta = 1.5f; cfg.sigma_min = 0.05f; cfg.sigma_max = 0.5f; cfg.input_channels = 4;
    use_handle* h = NULL;
    CHECK(use_create(&cfg, 0, &h));
    CHECK(use_load_weight_blob(h, argv[1]));

    float* win = (float*)malloc(sizeof(float) * n_fft);                     /* periodic Hann (model_wrapper.py:14-20) */
    for (int n = 0; n < n_fft; ++n) win[n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)n / (double)n_fft));
    float *d_wav = NULL, *d_out = NULL, *d_win = NULL; void *d_Y = NULL, *d_X = NULL;
    const size_t spec_bytes = (size_t)F * Tpad * 8;
    HIPCHECK(hipMalloc((void**)&d_wav, L * 4)); HIPCHECK(hipMalloc((void**)&d_out, L * 4)); HIPCHECK(hipMalloc((void**)&d_win, n_fft * 4));
    HIPCHECK(hipMalloc(&d_Y, spec_bytes)); HIPCHECK(hipMalloc(&d_X, spec_bytes));
    HIPCHECK(hipMemcpy(d_wav, wav, L * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_win, win, n_fft * 4, hipMemcpyHostToDevice));

    CHECK(use_stft_fwd(d_wav, d_Y, 1, (int)L, n_fft, hop, d_win, Tpad, factor, expo, NULL));
    CHECK(use_plan(h, 1, Tpad));
    use_sampler_config sc;
    memset(&sc, 0, sizeof sc);
    sc.N = N; sc.predictor = USE_PRED_REVERSE_DIFFUSION; sc.corrector = USE_CORR_LANGEVIN; sc.corrector_steps = 1;
    sc.snr = 0.5f; sc.t_eps = 3e-2f; sc.use_graph = 1;
    CHECK(use_set_sampler(h, &sc));
    CHECK(use_sample(h, d_Y, NULL, seed, d_X, NULL));                      /* device Philox noise */
    CHECK(use_istft_back(d_X, d_out, 1, (int)L, n_fft, hop, d_win, Tpad, factor, expo, NULL));
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(wav, d_out, L * 4, hipMemcpyDeviceToHost));
    CHECK(use_wav_write(argv[3], wav, L, 1, sr, USE_WAV_PCM16));
    printf("%s: %lld samples at %d Hz, %d frames (T' = %d), %d-step PC sampler -> %s\n", argv[2], (long long)L, sr, T, Tpad, N, argv[3]);
    use_destroy(h); use_free(wav); free(win);
    hipFree(d_wav); hipFree(d_out); hipFree(d_win); hipFree(d_Y); hipFree(d_X);
    return 0;
}
