// logmel — VGGish audio front-end on the device: framed STFT magnitude -> mel filterbank -> log, in float64.
// Replaces mel_features.log_mel_spectrogram / stft_magnitude / frame of the reference
// (contrastive_video_textures/utils/mel_features.py:21-92, 176-205; called from vggish_utils.py:27-69, which the
// reference runs in NumPy float64 once per video) and the example framing of vggish_utils.py:60-68.
//
// Once-per-video work (a 60 s clip is 6,000 frames x 257 bins x 400 samples = 1.2 GFLOP fp64), so the kernel is
// written for accuracy and one pass over the waveform, not for the FFT's operation count: one workgroup per STFT
// frame keeps the windowed frame, the twiddle table exp(-2 pi i k / fft_len) and the magnitudes in LDS; every thread
// sums its bins directly (a length-`win` dot product with exact table indices (b*n) mod fft_len), then the first
// n_mel threads apply the filterbank column by column.  Window and filterbank arrive as float64 tables built on the
// host with the reference's formulas (mel_features.py:41-43 periodic Hann, :117-173 HTK mel matrix), so the only
// differences to NumPy are summation order and the twiddles' last bit: ~1e-13 relative on the log-mel values.
#include "avt_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxFft = 2048;

struct LArgs {
  const void* wave;
  int wave_is_f64;
  int64_t n_frames;
  const double* window;
  int win, hop, fft_len, n_mel;
  const double* melmat;  // [fft_len/2+1, n_mel]
  double log_offset;
  double* out;  // [n_frames, n_mel]
};

__global__ __launch_bounds__(kThreads) void logmel_kernel(LArgs a) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* xw = sm;                   // [win]
  double* tc = xw + a.win;           // [fft_len] cos
  double* ts = tc + a.fft_len;       // [fft_len] sin
  double* spec = ts + a.fft_len;     // [fft_len/2+1]
  const int tid = threadIdx.x;
  const int64_t f = blockIdx.x;
  const int64_t base = f * a.hop;
  for (int n = tid; n < a.win; n += kThreads) {
    const double v = a.wave_is_f64 ? static_cast<const double*>(a.wave)[base + n]
                                   : (double)static_cast<const float*>(a.wave)[base + n];
    xw[n] = v * a.window[n];
  }
  for (int k = tid; k < a.fft_len; k += kThreads) {
    double s, c;
    sincospi(2.0 * (double)k / (double)a.fft_len, &s, &c);  // exact argument: fft_len is a power of two
    tc[k] = c;
    ts[k] = s;
  }
  __syncthreads();
  const int nb = a.fft_len / 2 + 1, mask = a.fft_len - 1;
  for (int b = tid; b < nb; b += kThreads) {
    double re = 0.0, im = 0.0;
    int idx = 0;
    for (int n = 0; n < a.win; ++n) {
      const double x = xw[n];
      re += x * tc[idx];
      im -= x * ts[idx];
      idx = (idx + b) & mask;
    }
    spec[b] = sqrt(re * re + im * im);
  }
  __syncthreads();
  for (int m = tid; m < a.n_mel; m += kThreads) {
    double acc = 0.0;
    for (int b = 0; b < nb; ++b) acc += spec[b] * a.melmat[(int64_t)b * a.n_mel + m];
    a.out[f * a.n_mel + m] = log(acc + a.log_offset);
  }
}

__global__ __launch_bounds__(kThreads) void examples_kernel(const double* __restrict__ logmel, int n_mel, int ex_len,
                                                            int ex_hop, int64_t total, float* __restrict__ out) {
  const int64_t per = (int64_t)ex_len * n_mel;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
    const int64_t e = i / per, r = i - e * per;  // r = row * n_mel + m inside the example
    out[i] = (float)logmel[e * ex_hop * n_mel + r];
  }
}

}  // namespace

extern "C" int avt_logmel_f64(const void* wave, int wave_is_f64, int64_t n_samples, const double* window, int win,
                              int hop, int fft_len, const double* melmat, int n_mel, double log_offset, double* logmel,
                              void* stream) {
  AVT_REQUIRE(n_samples >= 0 && win > 0 && hop > 0 && n_mel > 0, "avt_logmel_f64: bad sizes");
  AVT_REQUIRE(fft_len >= win && fft_len <= kMaxFft && (fft_len & (fft_len - 1)) == 0,
              "avt_logmel_f64: fft_len=%d must be a power of two in [win=%d, %d]", fft_len, win, kMaxFft);
  if (n_samples < win) return AVT_OK;  // no complete frame (mel_features.frame drops the tail)
  AVT_REQUIRE(wave && window && melmat && logmel, "avt_logmel_f64: NULL pointer");
  LArgs a;
  a.wave = wave;
  a.wave_is_f64 = wave_is_f64;
  a.n_frames = 1 + (n_samples - win) / hop;
  a.window = window;
  a.win = win;
  a.hop = hop;
  a.fft_len = fft_len;
  a.n_mel = n_mel;
  a.melmat = melmat;
  a.log_offset = log_offset;
  a.out = logmel;
  AVT_REQUIRE(a.n_frames < (1ll << 31), "avt_logmel_f64: too many frames");
  const size_t lds = (size_t)(win + 2 * fft_len + fft_len / 2 + 1) * sizeof(double);
  hipLaunchKernelGGL(logmel_kernel, dim3((unsigned)a.n_frames), dim3(kThreads), lds, static_cast<hipStream_t>(stream), a);
  return avt::check_launch("avt_logmel_f64");
}

extern "C" int avt_logmel_examples_f32(const double* logmel, int64_t n_frames, int n_mel, int ex_len, int ex_hop,
                                       float* out, void* stream) {
  AVT_REQUIRE(n_frames >= 0 && n_mel > 0 && ex_len > 0 && ex_hop > 0, "avt_logmel_examples_f32: bad sizes");
  if (n_frames < ex_len) return AVT_OK;  // no complete example
  AVT_REQUIRE(logmel && out, "avt_logmel_examples_f32: NULL pointer");
  const int64_t n_ex = 1 + (n_frames - ex_len) / ex_hop;
  const int64_t total = n_ex * ex_len * n_mel;
  const int64_t blocks = (total + kThreads - 1) / kThreads;
  hipLaunchKernelGGL(examples_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), logmel, n_mel, ex_len, ex_hop, total, out);
  return avt::check_launch("avt_logmel_examples_f32");
}
