// Wire formats either side of the sampling path, host code: WAV files in and out and the 24 kHz FFT resampling of
// the reference's inference loader, so that the C-ABI library can go file -> enhanced file without Python.
//
//   use_wav_read / use_load_utterance  <- LoadWavDataset.__getitem__        src/data/components/loadwav_dataset.py:90-120
//        sf.read (libsndfile: integer PCM scaled by 2^-(bits-1), float as stored) -> first channel ->
//        librosa.resample(res_type="fft") == scipy.signal.resample(x, ceil(len * target / sr)) -> x / max|x| * 0.8, all in
//        float64, then float32.  librosa / soundfile are third-party packages absent from the reference tree and from this image
//        (requirements.txt pins librosa==0.10.2, soundfile==0.12.1); the resampler restates scipy.signal.resample's published
//        algorithm (spectrum truncation / zero padding with the Nyquist bin split, scipy 1.15) and is pinned against scipy in
//        tests/test_io.py.
//   use_wav_write                      <- sf.write(path, enhanced, sampling_rate)                  src/models/SGMSE_module.py:80
//        soundfile's default WAV subtype is PCM_16 whatever the array dtype: round(x * 32767) (libsndfile's normalised float ->
//        short conversion); values beyond full scale are clipped here (libsndfile wraps them unless clipping is switched on).
//
// The transforms are double-precision Bluestein FFTs (any length: utterance lengths are arbitrary) over a radix-2 kernel.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <complex>
#include <string>
#include <vector>
#include "../../include/use_hip.h"

int use_set_error(int code, const char* msg);   // use_engine.cpp

namespace {

typedef std::complex<double> cd;

int failf(int code, const char* fmt, const char* a, long b = 0) {
    char buf[1024];
    snprintf(buf, sizeof buf, fmt, a, b);
    return use_set_error(code, buf);
}

// in-place radix-2 DIT FFT, n a power of two; sign -1: forward
void fft_pow2(std::vector<cd>& a, int sign) {
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const size_t half = len >> 1;
        std::vector<cd> w(half);                                  // exact twiddles per stage (no recurrence drift)
        for (size_t k = 0; k < half; ++k) {
            const double ang = sign * 2.0 * M_PI * (double)k / (double)len;
            w[k] = cd(cos(ang), sin(ang));
        }
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < half; ++k) {
                const cd u = a[i + k], v = a[i + k + half] * w[k];
                a[i + k] = u + v; a[i + k + half] = u - v;
            }
    }
}

// DFT of any length: X[k] = sum_n x[n] exp(sign * 2 pi i k n / N)  (Bluestein: k n = (k^2 + n^2 - (k - n)^2) / 2)
void dft(std::vector<cd>& x, int sign) {
    const size_t N = x.size();
    if (N <= 1) return;
    if ((N & (N - 1)) == 0) { fft_pow2(x, sign); return; }
    size_t M = 1; while (M < 2 * N - 1) M <<= 1;
    std::vector<cd> chirp(N);
    for (size_t k = 0; k < N; ++k) {
        const uint64_t k2 = ((uint64_t)k * (uint64_t)k) % (2 * (uint64_t)N);    // angle reduced exactly
        const double ang = sign * M_PI * (double)k2 / (double)N;
        chirp[k] = cd(cos(ang), sin(ang));
    }
    std::vector<cd> a(M, cd(0, 0)), b(M, cd(0, 0));
    for (size_t n = 0; n < N; ++n) a[n] = x[n] * chirp[n];
    b[0] = std::conj(chirp[0]);
    for (size_t n = 1; n < N; ++n) b[n] = b[M - n] = std::conj(chirp[n]);
    fft_pow2(a, -1); fft_pow2(b, -1);
    for (size_t i = 0; i < M; ++i) a[i] *= b[i];
    fft_pow2(a, +1);
    const double inv = 1.0 / (double)M;
    for (size_t k = 0; k < N; ++k) x[k] = a[k] * inv * chirp[k];
}

// scipy.signal.resample(x, num) for real x (time domain, no window)
void resample_fft(const double* x, int64_t Nx, int64_t num, double* y) {
    if (num == Nx) { memcpy(y, x, (size_t)Nx * sizeof(double)); return; }
    std::vector<cd> X((size_t)Nx);
    for (int64_t i = 0; i < Nx; ++i) X[(size_t)i] = cd(x[i], 0.0);
    dft(X, -1);                                                       // rfft = first Nx/2+1 bins
    const int64_t N = num < Nx ? num : Nx, nyq = N / 2 + 1, nh = num / 2 + 1;
    std::vector<cd> Y((size_t)nh, cd(0, 0));
    for (int64_t k = 0; k < nyq && k < nh; ++k) Y[(size_t)k] = X[(size_t)k];
    if (N % 2 == 0) {                                                 // the Nyquist bin of the shorter grid is shared by +-N/2
        if (num < Nx) Y[(size_t)(N / 2)] *= 2.0;
        else          Y[(size_t)(N / 2)] *= 0.5;
    }
    // irfft(Y, num): Hermitian extension (the imaginary parts of DC and of an even grid's Nyquist bin do not contribute)
    std::vector<cd> Z((size_t)num, cd(0, 0));
    Z[0] = cd(Y[0].real(), 0.0);
    for (int64_t k = 1; k < nh; ++k) {
        if (2 * k == num) { Z[(size_t)k] = cd(Y[(size_t)k].real(), 0.0); continue; }
        Z[(size_t)k] = Y[(size_t)k];
        Z[(size_t)(num - k)] = std::conj(Y[(size_t)k]);
    }
    dft(Z, +1);
    const double sc = (1.0 / (double)num) * ((double)num / (double)Nx);
    for (int64_t i = 0; i < num; ++i) y[i] = Z[(size_t)i].real() * sc;
}

uint32_t rd32(const unsigned char* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
uint16_t rd16(const unsigned char* p) { return (uint16_t)(p[0] | p[1] << 8); }

}  // namespace

extern "C" {

void use_free(void* p) { free(p); }

int use_wav_read(const char* path, double** samples, int64_t* frames, int* channels, int* sample_rate) {
    if (!path || !samples || !frames || !channels || !sample_rate) return use_set_error(USE_E_INVALID, "use_wav_read: null argument");
    FILE* f = fopen(path, "rb");
    if (!f) return failf(USE_E_INVALID, "cannot open '%s'", path);
    std::vector<unsigned char> buf;
    {
        fseek(f, 0, SEEK_END); const long sz = ftell(f); fseek(f, 0, SEEK_SET);
        if (sz < 12) { fclose(f); return failf(USE_E_INVALID, "'%s' is not a RIFF/WAVE file", path); }
        buf.resize((size_t)sz);
        const size_t got = fread(buf.data(), 1, (size_t)sz, f);
        fclose(f);
        if (got != (size_t)sz) return failf(USE_E_INVALID, "short read on '%s'", path);
    }
    if (memcmp(buf.data(), "RIFF", 4) || memcmp(buf.data() + 8, "WAVE", 4)) return failf(USE_E_INVALID, "'%s' is not a RIFF/WAVE file", path);
    int fmt = 0, ch = 0, bits = 0, align = 0; uint32_t sr = 0;
    const unsigned char* data = nullptr; size_t data_len = 0;
    for (size_t pos = 12; pos + 8 <= buf.size();) {
        const unsigned char* c = buf.data() + pos;
        size_t len = rd32(c + 4);
        if (pos + 8 + len > buf.size()) len = buf.size() - pos - 8;   // truncated last chunk (streamed writers leave 0 / 0xffffffff)
        if (!memcmp(c, "fmt ", 4) && len >= 16) {
            fmt = rd16(c + 8); ch = rd16(c + 10); sr = rd32(c + 12); align = rd16(c + 20); bits = rd16(c + 22);
            if (fmt == 0xFFFE && len >= 26) fmt = rd16(c + 8 + 24);   // WAVE_FORMAT_EXTENSIBLE: first two bytes of the sub-format GUID
        } else if (!memcmp(c, "data", 4)) {
            data = c + 8; data_len = len;
            break;
        }
        pos += 8 + len + (len & 1);
    }
    if (!data || ch < 1 || bits < 8) return failf(USE_E_INVALID, "'%s': no fmt/data chunk", path);
    const int bps = bits / 8;
    if (align < bps * ch) align = bps * ch;
    if (!((fmt == 1 && (bps >= 1 && bps <= 4)) || (fmt == 3 && (bps == 4 || bps == 8))))
        return failf(USE_E_INVALID, "'%s': unsupported sample format (tag %ld)", path, (long)fmt * 100 + bits);
    const int64_t n = (int64_t)(data_len / (size_t)align);
    double* out = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1) * (size_t)ch);
    if (!out) return use_set_error(USE_E_NOMEM, "use_wav_read: out of memory");
    for (int64_t i = 0; i < n; ++i)
        for (int c = 0; c < ch; ++c) {
            const unsigned char* q = data + (size_t)i * (size_t)align + (size_t)c * (size_t)bps;
            double v;
            if (fmt == 3) {
                if (bps == 4) { float t; memcpy(&t, q, 4); v = (double)t; } else { memcpy(&v, q, 8); }
            } else if (bps == 1) v = ((double)q[0] - 128.0) / 128.0;
            else if (bps == 2) v = (double)(int16_t)rd16(q) / 32768.0;
            else if (bps == 3) v = (double)((int32_t)((uint32_t)q[0] << 8 | (uint32_t)q[1] << 16 | (uint32_t)q[2] << 24) >> 8) / 8388608.0;
            else v = (double)(int32_t)rd32(q) / 2147483648.0;
            out[(size_t)i * (size_t)ch + (size_t)c] = v;
        }
    *samples = out; *frames = n; *channels = ch; *sample_rate = (int)sr;
    return USE_OK;
}

int use_wav_write(const char* path, const float* samples, int64_t frames, int channels, int sample_rate, int subtype) {
    if (!path || (!samples && frames > 0) || frames < 0 || channels < 1 || sample_rate < 1 || (subtype != USE_WAV_PCM16 && subtype != USE_WAV_FLOAT32))
        return use_set_error(USE_E_INVALID, "use_wav_write: bad argument");
    const int bps = subtype == USE_WAV_PCM16 ? 2 : 4;
    const uint64_t data_len = (uint64_t)frames * (uint64_t)channels * (uint64_t)bps;
    if (data_len > 0xFFFFFF00ull) return use_set_error(USE_E_INVALID, "use_wav_write: more than 4 GiB of samples");
    FILE* f = fopen(path, "wb");
    if (!f) return failf(USE_E_INVALID, "cannot create '%s'", path);
    unsigned char h[44];
    auto w32 = [](unsigned char* p, uint32_t v) { p[0] = v & 255; p[1] = v >> 8 & 255; p[2] = v >> 16 & 255; p[3] = v >> 24 & 255; };
    auto w16 = [](unsigned char* p, uint16_t v) { p[0] = v & 255; p[1] = v >> 8 & 255; };
    memcpy(h, "RIFF", 4); w32(h + 4, (uint32_t)(36 + data_len)); memcpy(h + 8, "WAVEfmt ", 8); w32(h + 16, 16);
    w16(h + 20, subtype == USE_WAV_PCM16 ? 1 : 3); w16(h + 22, (uint16_t)channels); w32(h + 24, (uint32_t)sample_rate);
    w32(h + 28, (uint32_t)sample_rate * (uint32_t)channels * (uint32_t)bps); w16(h + 32, (uint16_t)(channels * bps)); w16(h + 34, (uint16_t)(bps * 8));
    memcpy(h + 36, "data", 4); w32(h + 40, (uint32_t)data_len);
    bool ok = fwrite(h, 1, 44, f) == 44;
    const size_t n = (size_t)frames * (size_t)channels;
    if (subtype == USE_WAV_FLOAT32) {
        ok = ok && fwrite(samples, 4, n, f) == n;                     // little-endian host (gfx950 hosts are x86-64)
    } else {
        std::vector<int16_t> q(n);
        for (size_t i = 0; i < n; ++i) {
            double v = nearbyint((double)samples[i] * 32767.0);
            if (!(v == v)) v = 0.0;
            q[i] = (int16_t)(v > 32767.0 ? 32767.0 : v < -32768.0 ? -32768.0 : v);
        }
        ok = ok && fwrite(q.data(), 2, n, f) == n;
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? USE_OK : failf(USE_E_INVALID, "write to '%s' failed", path);
}

int use_resample_fft(const double* x, int64_t n, int64_t num, double* y) {
    if (!x || !y || n < 1 || num < 1) return use_set_error(USE_E_INVALID, "use_resample_fft: bad argument");
    resample_fft(x, n, num, y);
    return USE_OK;
}

int use_load_utterance(const char* path, int target_rate, int normalize, float** wav, int64_t* length, int* sample_rate) {
    if (!wav || !length || !sample_rate) return use_set_error(USE_E_INVALID, "use_load_utterance: null argument");
    double* raw = nullptr; int64_t frames = 0; int ch = 0, sr = 0;
    const int rc = use_wav_read(path, &raw, &frames, &ch, &sr);
    if (rc) return rc;
    if (frames < 1) { free(raw); return failf(USE_E_INVALID, "'%s' holds no samples", path); }
    std::vector<double> x((size_t)frames);
    for (int64_t i = 0; i < frames; ++i) x[(size_t)i] = raw[(size_t)i * (size_t)ch];        // first channel (loadwav_dataset.py:93-94)
    free(raw);
    if (target_rate > 0 && target_rate != sr) {                                             // loadwav_dataset.py:95-98
        // librosa.resample: ratio = float(target_sr) / orig_sr first, then int(ceil(len * ratio)) - for 44.1 kHz-family rates and
        // lengths that are multiples of 147 the rounded ratio gives one sample more than ceil(len * target / sr) would
        const double ratio = (double)target_rate / (double)sr;
        const int64_t num = (int64_t)ceil((double)frames * ratio);
        std::vector<double> y((size_t)num);
        resample_fft(x.data(), frames, num, y.data());
        x.swap(y);
        sr = target_rate;
    }
    if (normalize) {                                                                        // loadwav_dataset.py:99-100
        double mx = 0.0;
        for (double v : x) mx = fabs(v) > mx ? fabs(v) : mx;
        for (double& v : x) v = v / mx * 0.8;                                                // (a silent file gives NaN, as in the reference)
    }
    float* out = (float*)malloc(sizeof(float) * x.size());
    if (!out) return use_set_error(USE_E_NOMEM, "use_load_utterance: out of memory");
    for (size_t i = 0; i < x.size(); ++i) out[i] = (float)x[i];
    *wav = out; *length = (int64_t)x.size(); *sample_rate = sr;
    return USE_OK;
}

}  // extern "C"
