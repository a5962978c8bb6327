/* noise_hip.h - C ABI of libnoise_hip.so: MI355X (gfx950) kernels for the speech-enhancement hot path.
 *
 * The reference (vinhng10/noise) has no native boundary: its hot path is a sequence of ATen ops
 * called from Python.  Each entry point below replaces the ATen op sequence at the cited reference
 * lines (paths relative to the reference root).  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (plain pointers and sizes only);
 *     the library never allocates, frees or retains memory; `scratch` arguments are caller-owned
 *     workspaces whose minimum sizes are stated per function;
 *   - launches are asynchronous on `stream` (a hipStream_t passed as void*);
 *   - return 0 on success; otherwise nothing was launched and noise_last_error() describes why;
 *   - activations are NHWC fp32; spectra are [B][F][T] fp32; waveforms [B][L] fp32;
 *   - "lazy source" (y, coef, relu6): a = act(coef[c]*y + coef[C+c]) with
 *     coef = [scale|shift|mean|invstd|mean_lo] (5*C floats; may be NULL: a = y);
 *     "lazy gradient" (dz, y, coef3): dy = s*(dz - k1) + a2*(y - mean) with
 *     coef3 = [s|k1_hi|k1_lo|a2|mean_hi|mean_lo] (6*C floats; may be NULL: dy = dz).
 */
#ifndef NOISE_HIP_H
#define NOISE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library ------------------------------------------------------------------------------- */
const char* noise_last_error(void);   /* text of the last failure on the calling thread */
int noise_version(void);

/* ---- STFT / iSTFT: train/models.py:460-475 (STFT.forward), :477-495 (STFT.inverse) ---------- */
int noise_stft_rpad(int n_fft);       /* row count of the interleaved, transposed basis layout */
/* wave [B][L] -> logmag, phase [B][F][T] (phase optional: NULL skips it), T = L/hop + 1; absmax [B] (optional) = max |logmag| per clip
 * (models.py:614-620).  basisT [n_fft][rpad]: column 2f = Re row f, 2f+1 = Im row f of forward_basis. */
int noise_stft_fwd(const float* wave, const float* basisT, float* logmag, float* phase, float* absmax,
                   int B, int L, int n_fft, int hop, void* stream);
/* logmag, phase [B][F][T] -> wave_out [B][hop*(T-1)], clamped to [-1,1]; inv_basis [2F][n_fft] and
 * window_sum as registered by the reference module; vad_logit [B] (optional) applies the inference
 * gate wave *= sigmoid(1e4*vad) (models.py:754). */
int noise_istft_fwd(const float* logmag, const float* phase, const float* inv_basis, const float* window_sum,
                    const float* vad_logit, float* wave_out, int B, int T, int n_fft, int hop, void* stream);
/* autograd of models.py:477-495 w.r.t. log_magnitude; inv_basisT in the noise_stft_fwd basis layout */
int noise_istft_bwd(const float* d_wave, const float* wave_out, const float* logmag, const float* phase,
                    const float* inv_basisT, const float* window_sum, float* d_logmag, int accumulate,
                    int B, int T, int n_fft, int hop, void* stream);
/* Symmetric-window variants of the three entry points above (same reference lines).  When the window
 * satisfies w[n] = w[n_fft - n] (every scipy periodic window the reference can build, models.py:433-447) the
 * Re rows of both bases are even and the Im rows odd about n_fft/2, so frames are folded
 * (x[k] +- x[n_fft - k]) and only basis rows 0..n_fft/2 are multiplied: half the arithmetic, results equal up
 * to summation order.  basisF / inv_basisF [n_fft/2 + 1][cols], cols = noise_stft_fold_cols(n_fft):
 * column f = Re row f, column cols/2 + f = Im row f of the basis, taken at sample k.  noise_istft_fwd_sym reads
 * the reference's inverse_basis buffer as is.  The caller checks the symmetry of its bases. */
int noise_stft_fold_cols(int n_fft);
int noise_stft_fwd_sym(const float* wave, const float* basisF, float* logmag, float* phase, float* absmax,
                       int B, int L, int n_fft, int hop, void* stream);
int noise_istft_fwd_sym(const float* logmag, const float* phase, const float* inv_basis, const float* window_sum,
                        const float* vad_logit, float* wave_out, int B, int T, int n_fft, int hop, void* stream);
int noise_istft_bwd_sym(const float* d_wave, const float* wave_out, const float* logmag, const float* phase,
                        const float* inv_basisF, const float* window_sum, float* d_logmag, int accumulate,
                        int B, int T, int n_fft, int hop, void* stream);

/* FFT variants of the same three operations (same reference lines), for bases that are a window times the real DFT, as
 * the reference registers them (models.py:413-458): forward_basis[f][n] = w[n] (cos, -sin)(2 pi f n / N),
 * inverse_basis[f][n] = c_f ws[n] (cos, -sin)(2 pi f n / N) / N, c = 1 for the DC / Nyquist rows and 2 otherwise.  The
 * caller checks that structure and passes window = w (forward) or syn_window = ws (inverse and adjoint), each [n_fft],
 * plus the twiddle tables tw1 [n_fft/2][2] = e^{-2 pi i j / (n_fft/2)} and tw2 [n_fft/2 + 1][2] = e^{-2 pi i k / n_fft}
 * as (re, im) pairs.  A frame is one complex FFT of n_fft/2 points done in two register passes through LDS plus a split
 * step; about 1/15 of the arithmetic of the dense products.  noise_fft_supported(n_fft) != 0 for the sizes served
 * (320 = 2*16*10 and 512 = 2*16*16); results agree with the dense entry points to fp32 rounding. */
int noise_fft_supported(int n_fft);
int noise_stft_fwd_fft(const float* wave, const float* window, const float* tw1, const float* tw2, float* logmag,
                       float* phase, float* absmax, int B, int L, int n_fft, int hop, int fast_math, void* stream);
/* noise_stft_fwd_fft with the per-clip normaliser absmax (models.py:614-620) evaluated in the accumulation order of
 * noise_stft_fwd, i.e. of the reference's conv1d (models.py:465-467): the bins that can attain the max are recomputed as
 * dense k-ordered dot products against basisT (layout of noise_stft_fwd).  The network input of the training step. */
int noise_stft_fwd_fft_ref(const float* wave, const float* window, const float* tw1, const float* tw2, const float* basisT,
                           float* logmag, float* phase, float* absmax, int B, int L, int n_fft, int hop, void* stream);
int noise_istft_fwd_fft(const float* logmag, const float* phase, const float* syn_window, const float* tw1, const float* tw2,
                        const float* window_sum, const float* vad_logit, float* wave_out, int B, int T, int n_fft, int hop,
                        int fast_math, void* stream);
int noise_istft_bwd_fft(const float* d_wave, const float* wave_out, const float* logmag, const float* phase,
                        const float* syn_window, const float* tw1, const float* tw2, const float* window_sum,
                        float* d_logmag, int accumulate, int B, int T, int n_fft, int hop, void* stream);

/* Projection of a transposed inverted-residual block without its depthwise output in memory (models.py:299-345,
 * :385-396): out (B, 2Hi-1, 2Wi-1, N) raw = act_dw(bn_dw(ConvTranspose2d_dw3x3s2(act_in(bn_in(in))))) x W^T with in
 * (B, Hi, Wi, K) raw, wdw (K, 1, 3, 3), W (N, K); *_coef = [scale(K) | shift(K) | ...] or NULL.  Same arithmetic and
 * summation order as noise_dw3x3_fwd (transposed) followed by noise_pw_fwd.  Served: (K, N) = (72, 16), (96, 24),
 * (192, 32), the decoder blocks of spectralv2.yaml.  stats_slab as noise_pw_fwd (<= 1024 rows). */
int noise_pw_fwd_virt_supported(int K, int N);
int noise_pw_fwd_virt(const float* in_y, const float* in_coef, int in_relu6, const float* wdw, const float* dw_coef,
                      int dw_relu6, const float* w, float* out, float* stats_slab, int* slab_rows, int B, int Hi, int Wi,
                      int K, int N, void* stream);
/* Backward of noise_pw_fwd_virt (fused as noise_pw_bwd): g_in (B, 2Hi-1, 2Wi-1, K) = gradient of the activated
 * depthwise output masked by the depthwise ReLU6; stats_slab rows [sum g | sum g*xhat] (2K floats) for the depthwise
 * BatchNorm backward; scratch rows of N*K floats with the per-workgroup weight gradient (*slab_rows rows of each, at
 * most noise_pw_bwd_max_rows(N, K)).  dz / y_out / coef3: lazy gradient of the projection output (coef3 NULL: dz is
 * dL/dy).  dw_coef: [scale | shift | mean | invstd | mean_lo] of the depthwise BatchNorm (required). */
int noise_pw_bwd_virt(const float* dz, const float* y_out, const float* coef3, const float* in_y, const float* in_coef,
                      int in_relu6, const float* wdw, const float* dw_coef, int dw_relu6, const float* w, float* g_in,
                      float* scratch, float* stats_slab, int* slab_rows, int B, int Hi, int Wi, int K, int N,
                      void* stream);

/* ---- loss of the waveform models (train/models.py:641-720) ------------------------------------------------------ */
/* One resolution of Base.multi_resolution_stft_loss (:700-720) with Base.stft_mag (:687-698): torch.stft semantics
 * (reflect-padded centred frames, T = 1 + L/hop, window of win_length samples centred in n_fft), magnitudes clamped at
 * sqrt(1e-7), loss[0] (+)= mean |log mag(clean) - log mag(enhanced)|; d_enhanced (+)= grad_scale * d loss / d enhanced.
 * window [n_fft] is the padded window; tw1 [n_fft/2][2] = e^{-2 pi i j/(n_fft/2)}, tw2 [n_fft/2+1][2] = e^{-2 pi i k/n_fft}.
 * Workspaces: frame_grad [B][T][win_length] floats (NULL together with d_enhanced: loss only), partial
 * noise_mrstft_partials() doubles.  Serves n_fft 512, 1024, 2048 (the sizes of the reference's configs). */
int noise_mrstft_supported(int n_fft);
int noise_mrstft_partials(int B, int L, int n_fft, int hop);
int noise_mrstft_term(const float* enhanced, const float* clean, const float* window, const float* tw1, const float* tw2,
                      float grad_scale, float* frame_grad, double* partial, double* loss, int accumulate_loss,
                      float* d_enhanced, int accumulate_grad, int B, int L, int n_fft, int hop, int win_length,
                      void* stream);
/* F.smooth_l1_loss(x, y, beta) as :657 calls it (mean): loss[0] (+)= value, d_x (+)= grad_scale * gradient (d_x may be
 * NULL); partial: noise_smooth_l1_blocks() doubles. */
int noise_smooth_l1_blocks(void);
int noise_smooth_l1(const float* x, const float* y, long n, float beta, float grad_scale, double* partial, double* loss,
                    int accumulate_loss, float* d_x, int accumulate_grad, void* stream);

/* ---- deferred folds of per-workgroup slabs (no counterpart in the reference: ATen reduces inside each op) ------ */
/* out[j] = sum over rows of slab[row*row_stride + j], j < ncols; with tr_k > 0 column j = k*tr_C + c goes to
 * out[c*tr_k + k] (the (C, k) layout of a depthwise weight).  noise_pw_bwd and noise_dw3x3_bwd called with dw = NULL
 * leave their weight-gradient slab unfolded; one noise_fold_jobs call (jobs: DEVICE array) folds them all. */
typedef struct {
  const float* slab;
  float* out;
  int rows, ncols, row_stride, tr_k, tr_C;
} noise_fold_job;
int noise_fold_jobs(const noise_fold_job* jobs, int njobs, int max_cols, void* stream);

/* ---- waveform augmentation of the training data module: train/data.py:439-461 (chain), :418-424 (what the clean
 * clip keeps), :186-209 (ImpulseResponse.apply), :30-57 (Cut), :133-157 (noise scaling and tiling, as the reference
 * restates audiomentations' AddBackgroundNoise); process/process.py:69-75 (add_pyreverb) -------------------------- */
/* Parameters of one (noisy, clean) pair, drawn by the caller. */
typedef struct {
  int src;        /* row of noisy_src / clean_src */
  int shift;      /* Shift(rollover=False): samples moved towards later times (negative: earlier), vacated ones zero */
  float gain;     /* Gain: amplitude ratio 10^(dB/20), noisy clip only */
  float snr_db;   /* AddBackgroundNoise */
  long noise_off; /* first sample of the noise segment in noise_pool */
  int noise_len;  /* its length, <= L (0: no noise); a shorter segment repeats periodically (data.py:150-154) */
  int ir;         /* impulse response of the pool, -1: ImpulseResponse not applied */
  int ir_row;     /* row of the `mixed` / spectrum / reverb workspaces this clip uses, -1 without impulse response */
  int cut;        /* Cut: first sample of the output window, cut + length <= L */
  int invert;     /* PolarityInversion, noisy clip only */
} noise_aug_item;
/* Uniformly partitioned FFT convolution (overlap-save, blocks of noise_fir_block() = 1024 samples, 2048-point real
 * transforms).  noise_fir_spectra: x [n_rows] rows x_stride floats apart, L samples each -> spec
 * [n_rows][noise_fir_blocks(L)][1024][2]; filter = 0: segment k = samples (k-1)*1024 .. (k+1)*1024 (zeros outside the
 * row), filter = 1: segment k = partition k followed by 1024 zeros.  tw1 [1024][2] = e^{-2 pi i j/1024}, tw2 [1025][2] =
 * e^{-2 pi i k/2048}.
 * noise_fir_apply: y[row][0..L) = (x * h)[0..L), the convolution truncated to the input length as data.py:191-192 and
 * process.py:70-73 truncate fftconvolve(mode="full"); row uses impulse response ir_of_row[row] of a pool of n_ir, whose
 * spectra (noise_fir_spectra with filter = 1) start at block ir_first_block[i] of hspec and span ir_nblk[i] blocks.
 * partial (optional) [n_rows][noise_fir_partials(L)] receives sums of squares of y. */
int noise_fir_block(void);
int noise_fir_blocks(long L);
int noise_fir_partials(long L);
int noise_fir_spectra(const float* x, long x_stride, int n_rows, long L, int filter, const float* tw1, const float* tw2,
                      float* spec, void* stream);
int noise_fir_apply(const float* xspec, int n_rows, long L, const int* ir_of_row, const float* hspec,
                    const long* ir_first_block, const int* ir_nblk, int n_ir, const float* tw1, const float* tw2, float* y,
                    long y_stride, double* partial, void* stream);
/* Gain, Shift, AddBackgroundNoise of every item (items: DEVICE array).  Items with ir < 0 leave through Cut,
 * PolarityInversion and Clip into out_noisy [n_items][length]; the others write the whole mixed clip to
 * mixed[ir_row][L].  The input chain reads noisy_src, the target reads clean_src (the same pointer for
 * VADNoiseDataset, data.py:541-562, whose two passes both start from the clean clip).  out_clean [n_items][length] =
 * Clip(Cut(Shift(Gain(clean)))) with clean_gain != 0 (VADNoiseDataset: the frozen pass keeps Gain and Clip), else
 * Cut(Shift(clean)) (NoiseDataset, data.py:410-427).  partial: n_items * noise_aug_chunks(L) * 3 doubles (kept for
 * noise_aug_finish). */
int noise_aug_chunks(long L);
int noise_aug_mix(const noise_aug_item* items, int n_items, const float* noisy_src, const float* clean_src,
                  long src_stride, long L, const float* noise_pool, long noise_pool_len, double* partial, float* mixed,
                  float* out_noisy, float* out_clean, int length, int clean_gain, void* stream);
/* data.py:194-208 after the convolution (RMS of the input restored, clip), then Cut, PolarityInversion, Clip, for the
 * n_rows convolved clips: reverb [n_rows][L] and rev_partial from noise_fir_apply, item_of_row [n_rows] their items. */
int noise_aug_finish(const noise_aug_item* items, const int* item_of_row, int n_rows, const float* reverb, long L,
                     const double* mix_partial, const double* rev_partial, float* out_noisy, int length, void* stream);

/* ---- SNR mixer: process/audiolib.py:176-216 (+ :219-248, :30-35, :18-19), :251-289 ---------- */
/* clean, noise [U][L] float64 (already length-matched); snr, level_draw [U]; outputs [U][L], level_out [U] */
int noise_snr_mix(const double* clean, const double* noise, const int* snr, const int* level_draw,
                  double* clean_out, double* noise_out, double* noisy_out, int* level_out,
                  int n_utt, long L, double target_level, double clipping_threshold, void* stream);
/* the same mix for the generation loop of process/process.py:240-256, which asks is_clipped (audiolib.py:18-19) of all
 * three signals right after mixing: peaks_out [U][4] (optional) = max |clean_out|, max |noise_out|, max |noisy_out|,
 * sum of noisy_out (a checksum); noisy_f32 [U][L] (optional) = float32 copy of noisy_out, the input of the STFT
 * features */
int noise_snr_mix_peaks(const double* clean, const double* noise, const int* snr, const int* level_draw,
                        double* clean_out, double* noise_out, double* noisy_out, int* level_out, double* peaks_out,
                        float* noisy_f32, int n_utt, long L, double target_level, double clipping_threshold,
                        void* stream);
int noise_activity(const double* audio, double* perc_active, int n_utt, long L, int fs,
                   double energy_thresh, double target_level, void* stream);

/* ---- U-Net layers: train/models.py:583-612 (Base._forward), :348-407, :299-345, :1151-1201 -- */
int noise_conv_max_blocks(void);      /* slab rows a conv/bn kernel may write (sizes the scratch buffers) */
/* depthwise 3x3, pad 1 (Conv2d :385-393 or ConvTranspose2d :299-345), input = lazy a (+ lazy b);
 * out raw [B][Ho][Wo][C]; stats_slab (optional) receives *slab_rows rows of [sum(C)|sumsq(C)] */
int noise_dw3x3_fwd(const float* a_y, const float* a_coef, int a_relu6, const float* b_y, const float* b_coef,
                    int b_relu6, const float* w, float* out, float* stats_slab, int* slab_rows, int transposed,
                    int B, int Hi, int Wi, int C, int sy, int sx, void* stream);
/* g_in [B][Hi][Wi][C], dw (C,1,3,3); scratch >= noise_conv_max_blocks()*11*C floats.
 * Backward epilogue (all optional), shared by noise_dw3x3_bwd / noise_last_convT_bwd / noise_pw_bwd:
 *   g_in += g_other (a second gradient of the same tensor, e.g. from a skip add); with epi_coef (the BatchNorm
 *   coefficients of source a's producer) g_in is masked by that producer's ReLU6 (epi_relu6) and the producer's
 *   BatchNorm-backward sums [sum dz | sum dz*xhat] are left in scratch for noise_bn_bwd_finalize
 *   (here: *slab_rows rows of 11*C floats, sums in columns [9C, 11C)). */
int noise_dw3x3_bwd(const float* dz, const float* y_out, const float* coef3, const float* a_y, const float* a_coef,
                    int a_relu6, const float* b_y, const float* b_coef, int b_relu6, const float* w, float* g_in,
                    float* dw, float* scratch, const float* g_other, const float* epi_coef, int epi_relu6,
                    int* slab_rows, int transposed, int B, int Hi, int Wi, int C, int sy, int sx, void* stream);
/* first Conv2d 1->C (models.py:1151-1159) fused with normalise (:614-622) and left/top pad (:589) */
int noise_first_conv_fwd(const float* logmag, const float* absmax, const float* w, float* out, float* stats_slab,
                         int* slab_rows, int B, int H, int W, int hp, int wp, float pad_value, int C, int sy,
                         int sx, void* stream);
/* w (optional): with the layer's weights the stride-2 path rebuilds y_out from the input instead of reading it */
int noise_first_conv_wgrad(const float* dz, const float* y_out, const float* coef3, const float* logmag,
                           const float* absmax, const float* w, float* dw, float* scratch, int B, int H, int W,
                           int hp, int wp, float pad_value, int C, int sy, int sx, void* stream);
/* last ConvTranspose2d C->1 with bias (models.py:1164-1171) fused with crop (:610) and the additive mask
 * enh = clamp(logmag/max + filt, -1, 1)*max (:762-763); either of enh / filt may be NULL */
int noise_last_convT_fwd(const float* a_y, const float* a_coef, int a_relu6, const float* b_y, const float* b_coef,
                         int b_relu6, const float* w, const float* bias, const float* logmag, const float* absmax,
                         float* enh, float* filt, int B, int Hi, int Wi, int C, int sy, int sx, int H, int W,
                         void* stream);
/* scratch >= noise_conv_max_blocks()*12*C floats (rows of 12*C; producer sums in columns [10C, 12C)) */
int noise_last_convT_bwd(const float* d_enh, const float* enh, const float* absmax, const float* a_y,
                         const float* a_coef, int a_relu6, const float* b_y, const float* b_coef, int b_relu6,
                         const float* w, float* g_in, float* dw, float* dbias, float* scratch, const float* g_other,
                         const float* epi_coef, int epi_relu6, int* slab_rows, int B, int Hi, int Wi, int C, int sy,
                         int sx, int H, int W, void* stream);
/* crop-from-the-end of a skip tensor before the decoder add (models.py:608, `skip[..., -h:, -w:]`):
 * dst [B][Hd][Wd][C] = src [B][Hs][Ws][C] at rows oy.., columns ox.. (raw values; the lazy coefficients carry over) */
int noise_crop_nhwc(const float* src, float* dst, int B, int Hs, int Ws, int C, int oy, int ox, int Hd, int Wd,
                    void* stream);
/* its adjoint (autograd of the slice at models.py:608 in a training step): g_src [B][Hs][Ws][C] = g_dst inside the
 * window (which ends at the last row and column), zero on the rows / columns that were cut off */
int noise_crop_nhwc_bwd(const float* g_dst, float* g_src, int B, int Hs, int Ws, int C, int oy, int ox, int Hd, int Wd,
                        void* stream);
/* pointwise convs (models.py:371-381, :395-396) as fp32 MFMA products: out [npix][N] = a [npix][K] x W[N][K]^T */
int noise_pw_fwd(const float* a_y, const float* a_coef, int a_relu6, const float* b_y, const float* b_coef,
                 int b_relu6, const float* w, float* out, float* stats_slab, int* slab_rows, long npix, int K, int N,
                 void* stream);
int noise_pw_dgrad(const float* dz, const float* y_out, const float* coef3, const float* w, const float* add,
                   float* g_in, long npix, int Cout, int Cin, void* stream);
int noise_pw_wgrad_max_blocks(void);
/* scratch >= noise_pw_wgrad_max_blocks()*Cout*Cin floats */
int noise_pw_wgrad(const float* dz, const float* y_out, const float* coef3, const float* a_y, const float* a_coef,
                   int a_relu6, const float* b_y, const float* b_coef, int b_relu6, float* dw, float* scratch,
                   long npix, int Cout, int Cin, void* stream);

int noise_pw_bwd_max_rows(int Cout, int Cin);   /* slab rows noise_pw_bwd may use for a Cout x Cin layer */
int noise_pw_bwd_reads_y(int Cout, int Cin);    /* 0: y_out is recomputed from the layer input (expanding layers), never read */
/* fused dgrad + wgrad of a pointwise layer; stats_slab (with epi_coef) receives *slab_rows rows of 2*Cin floats;
 * scratch >= noise_pw_bwd_max_rows(Cout, Cin)*Cout*Cin floats, stats_slab that many rows */
int noise_pw_bwd(const float* dz, const float* y_out, const float* coef3, const float* a_y, const float* a_coef,
                 int a_relu6, const float* b_y, const float* b_coef, int b_relu6, const float* w, float* g_in, float* dw,
                 float* scratch, const float* g_other, const float* epi_coef, int epi_relu6, float* stats_slab,
                 int* slab_rows, long npix, int Cout, int Cin, void* stream);

/* ---- block-fused bf16 inference (eval mode, BatchNorm folded): models.py:348-407, :299-345, :583-612 ------------ */
/* One inverted-residual block with stride 2 per launch: expand 1x1 + ReLU6 -> depthwise 3x3 (transposed != 0:
 * ConvTranspose2d) + ReLU6 -> project 1x1, the hidden tensor staying in LDS.  x [B][Hi][Wi][Cin] and the optional skip
 * [B][Hs][Ws][Cin] (added to x at rows soy.., columns sox..: the crop of models.py:608) are NHWC bf16, or fp32 when
 * in_f32; out [B][Ho][Wo][Cout] is bf16, or fp32 when out_f32 (Ho = (Hi-1)/2+1, transposed: 2*Hi-1).  Weights with
 * the BatchNorm scale folded in, zero padded, "half" = IEEE binary16 (inside a kernel every tile is a block input or
 * sits behind a ReLU6, so half loses no range and keeps 11 bits; only what leaves a kernel is rounded to bf16):
 * w1 [ceil16(Hid)][ceil32(Cin)] half, b1 [ceil16(Hid)] fp32, wd [9][ceil8(Hid)] half (tap-major), b2 [ceil8(Hid)] half,
 * w3 [ceil16(Cout)][ceil32(Hid)] half, b3 [ceil16(Cout)] fp32.  Served: the blocks of train/configs/spectralv2.yaml:77-84
 * (32->96->64 writes fp32 for the attention kernels, 64->192->32 reads fp32 from them); others are refused (use the
 * per-layer fp32 entry points). */
int noise_fused_ir_block(const void* x, const void* skip, int in_f32, int Hs, int Ws, int soy, int sox, const void* w1,
                         const float* b1, const void* wd, const void* b2, const void* w3, const float* b3, void* out,
                         int out_f32, int transposed, int B, int Hi, int Wi, int Cin, int Hid, int Cout, void* stream);

/* Head of the fused network: normalise (models.py:614-622) + left/top pad (:589) -> first Conv2d 1->32 s2 + BN + ReLU6
 * (:1151-1159; kept as skip tensor a0) -> depthwise 3x3 s1 + BN + ReLU6 -> 1x1 32->16 + BN (:385-396; skip tensor a1).
 * w0 [32][32] half: row c = folded tap 4q + t of channel c at k = 8q + 2t and again at k = 8q + 2t + 1 (q < 3, t < 4,
 * 4q + t <= 8; the kernel multiplies the half-rounded input and its rounding remainder), zeros elsewhere; wd [9][32],
 * b2 [32] half;
 * w3 [16][32] half.  a0 [B][H1][W1][32], a1 [B][H1][W1][16] bf16, H1 = (H + hp - 1)/2 + 1, W1 likewise. */
int noise_fused_head(const float* logmag, const float* absmax, const void* w0, const float* b0, const void* wd,
                     const void* b2, const void* w3, const float* b3, void* a0, void* a1, int B, int H, int W, int hp,
                     int wp, float pad_value, void* stream);
/* Tail of the fused network: s = d2 + a1 (skip, cropped) -> depthwise ConvTranspose2d 3x3 s1 + BN + ReLU6 -> 1x1
 * 16->32 + BN (models.py:1193-1201) -> + a0 (skip, cropped) -> last ConvTranspose2d 32->1 s2 with bias (:1164-1171) ->
 * crop (:610) -> enh = clamp(logmag/max + filt, -1, 1)*max (:762-763).  d2 [B][h][w][16] bf16; the skips are read at
 * their crop offsets (oy, ox) inside their own [Hs][Ws]; wd [9][16], b2 [16] half; w3 [32][32] half (columns >= 16
 * zero); wl [16][32] half (row = tap, column = channel); enh / filt [B][H][W] fp32 (either may be NULL). */
int noise_fused_tail(const void* d2, const void* a1, int Hs1, int Ws1, int oy1, int ox1, const void* a0, int Hs0,
                     int Ws0, int oy0, int ox0, const void* wd, const void* b2, const void* w3, const float* b3,
                     const void* wl, float bias, const float* logmag, const float* absmax, float* enh, float* filt,
                     int B, int h, int w, int H, int W, void* stream);

/* ---- BatchNorm2d in train / eval mode (every norm_layer of models.py:371-397, :1151-1159) ---- */
int noise_bn_finalize_fwd(const float* slab, int slab_rows, int C, long count, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, long long* num_batches_tracked, float momentum,
                          float eps, float* coef, void* stream);
int noise_bn_coef_eval(int C, const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* coef, void* stream);
/* dz may alias g1; scratch >= noise_conv_max_blocks()*2*C floats */
int noise_bn_bwd(const float* g1, const float* g2, const float* y, const float* coef, int relu6, const float* gamma,
                 float* dz, float* dgamma, float* dbeta, float* coef3, float* scratch, long npix, int C, void* stream);

/* noise_bn_bwd without a workspace argument.  It has no counterpart in the reference: it exists for callers that cannot
 * allocate device memory (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  The
 * partial slab lives in a 2 MiB static device array that each GPU holds until the process ends; C <= 256 (larger C
 * is refused).  Calls that overlap on one GPU share that array and race: issue them on one stream. */
int noise_bn_bwd_static(const float* g1, const float* g2, const float* y, const float* coef, int relu6,
                        const float* gamma, float* dz, float* dgamma, float* dbeta, float* coef3, long npix, int C,
                        void* stream);

/* the finalize half of noise_bn_bwd, for sums left by a consumer's backward epilogue */
int noise_bn_bwd_finalize(const float* slab, int slab_rows, int row_stride, const float* coef, const float* gamma,
                          float* dgamma, float* dbeta, float* coef3, long npix, int C, void* stream);

/* ---- bottleneck attention block: models.py:207-225, :60-89, :27-37, :102-112, vad head :602 --- */
int noise_attn_max_blocks(void);
int noise_attn_pre_fwd(const float* x_y, const float* x_coef, int x_relu6, const float* ln_w, const float* ln_b,
                       const float* wq, const float* wk, const float* wv, float* xhat0, float* rstd0, float* qkv,
                       long ntok, int n, int d_model, int heads, int dk, void* stream);
/* keep: optional uint8 [B][heads][n][n] keep-mask of the attention dropout (models.py:34); otherwise a
 * counter-based RNG keyed on (seed, element index) with drop probability p_drop */
int noise_attn_core_fwd(const float* qkv, float* o, float* ml, const unsigned char* keep, unsigned long long seed,
                        float p_drop, long ntok, int n, int d_model, int heads, int dk, void* stream);
/* the same core for the bf16 inference path: no dropout, half-precision operands on the matrix cores, fp32 scores and
 * accumulation, any clip length whose K / V fit LDS (about 2,400 tokens) */
int noise_attn_core_fwd_f16(const float* qkv, float* o, long ntok, int n, int d_model, int heads, int dk, void* stream);
int noise_attn_core_bwd(const float* qkv, const float* d_o, const float* ml, const unsigned char* keep,
                        unsigned long long seed, float p_drop, float* d_qkv, long ntok, int n, int d_model,
                        int heads, int dk, void* stream);
int noise_attn_post_fwd(const float* o, const float* xhat0, const float* ln0_w, const float* ln0_b, const float* fc,
                        const float* ln1_w, const float* ln1_b, const float* w1, const float* b1, const float* w2,
                        const float* b2, const float* ln2_w, const float* ln2_b, const float* vad_w,
                        const float* vad_b, float* xhat1, float* rstd1, float* hdn, float* xhat2, float* rstd2,
                        float* x3, float* vad, long ntok, int n, int d_model, int heads, int dk, void* stream);
/* grads = [dW2|dW1|dWfc|dln2_w|dln2_b|db2|db1|dln1_w|dln1_b|dvad_w|dvad_b]; scratch: noise_attn_max_blocks() rows.
 * grads = NULL (here and in noise_attn_pre_bwd) leaves the min(ceil(ntok/64), noise_attn_max_blocks()) slab rows of
 * that layout unfolded in scratch for noise_fold_jobs. */
int noise_attn_post_bwd(const float* g3, const float* d_vad, const float* o, const float* xhat1, const float* rstd1,
                        const float* hdn, const float* xhat2, const float* rstd2, const float* x3, const float* fc,
                        const float* ln1_w, const float* ln1_b, const float* w1, const float* w2, const float* ln2_w,
                        const float* vad_w, float* d_o, float* d_x1, float* grads, float* scratch, long ntok, int n,
                        int d_model, int heads, int dk, void* stream);
/* grads = [dWq|dWk|dWv|dln0_w|dln0_b] */
int noise_attn_pre_bwd(const float* d_qkv, const float* d_x1, const float* xhat0, const float* rstd0,
                       const float* ln_w, const float* ln_b, const float* wq, const float* wk, const float* wv,
                       float* d_x0, float* grads, float* scratch, long ntok, int n, int d_model, int heads, int dk,
                       void* stream);

/* ---- losses of BaseSpectral._shared_step: models.py:776-790 ---------------------------------- */
int noise_absmax(const float* x, float* out, int B, long n, void* stream);          /* models.py:614-620 */
int noise_loss_max_blocks(void);
int noise_spectral_l1(const float* enh, const float* clean_logmag, float* d_enh, double* partial, int* n_partial,
                      long n, void* stream);                                          /* :777 */
int noise_vad_label(const float* clean, float* label, int B, int L, int min_active, void* stream);   /* :782 */
/* l1_loss((B,1,1,L),(B,L)) broadcasts to all (i,j) clip pairs (:786-787); partial: ceil(L/2) doubles */
int noise_wave_pair_l1(const float* wave, const float* clean, float* d_wave, double* partial, int* n_partial,
                       int B, int L, void* stream);
/* out4 = [loss, vad_loss, waveform_loss, spectral_loss] (:790-797); d_vad = dBCE/dlogit (:780-783) */
int noise_loss_finalize(const double* spec_partial, int n_spec, long spec_count, const double* wave_partial,
                        int n_wave, int B, int L, const float* vad, const float* label, float* d_vad, float* out4,
                        void* stream);

/* ---- optimizer: torch.optim.AdamW as configured by train/main.py:13-14, spectralv2.yaml:99-111 */
int noise_adamw_step(float* p, const float* g, float* m, float* v, long long* step, long n, float lr, float beta1,
                     float beta2, float eps, float weight_decay, float grad_scale, void* stream);

/* ---- once-for-all spectral U-Net forward (train/layers.py:496-668): every width, kernel size and depth is a run-time
 * choice, so these kernels are shape-generic (NHWC fp32, channel counts multiples of 4 up to any width).  Convolutions
 * write their raw output plus per-workgroup partial sums of y and y*y (`slab`: noise_ofa_max_rows() rows of 2C floats,
 * `*rows` = rows written), which noise_bn_finalize_fwd / noise_bn_coef_eval turn into the [scale|shift] the consumer
 * applies while loading (DynamicBatchNorm2d on the leading channels, layers.py:284-308). */
int noise_ofa_max_rows(void);
/* first layer (layers.py:543-550 through :253-272): 1 -> C channels, 3x3, stride 2, padding 1, on x [B][H][W] padded with
 * hp zero rows at the top and wp zero columns at the left (layers.py:646); w = Conv2d weight [Cmax][1][3][3] */
int noise_ofa_first_conv(const float* x, const float* w, float* out, float* slab, int* rows, int B, int H, int W, int hp,
                         int wp, int C, void* stream);
/* DynamicSeparableConv2d.get_active_filter (layers.py:162-196): centre k x k of the [Cmax][1][kmax][kmax] depthwise
 * weight, passed through the kmax->k1 matrix m1 and the k1->k2 matrix m2 (null: no such step) while the current size
 * exceeds k; written tap-major, wt[k*k][C]; flip != 0 reverses the taps, which is the form noise_ofa_dw wants for a
 * TRANSPOSED stride-1 layer (conv_transpose2d at stride 1 = conv2d with the flipped filter) */
int noise_ofa_dw_filter(const float* weight, int kmax, const float* m1, int k1, const float* m2, int k2, int k, int C,
                        int flip, float* wt, void* stream);
/* depthwise k x k conv2d / conv_transpose2d, stride 1 or 2, padding k/2 (layers.py:198-215) of the lazily normalised
 * source with the filter from noise_ofa_dw_filter (flipped there when transposed != 0 and stride == 1) */
int noise_ofa_dw(const float* src_y, const float* src_coef, int src_relu6, const float* wt, float* out, float* slab,
                 int* rows, int B, int Hi, int Wi, int C, int k, int stride, int transposed, void* stream);
/* DynamicPointConv2d with kernel 1 (layers.py:253-272): out [npix][N] = a [npix][K] x W[:N, :K]^T, W row stride ldw */
int noise_ofa_pw(const float* src_y, const float* src_coef, int src_relu6, const float* W, int ldw, float* out,
                 float* slab, int* rows, long npix, int K, int N, void* stream);
/* block output (layers.py:454-458, :662): out [B][Ho][Wo][C] = act(scale*y + shift) cropped from the end of y
 * [B][Hy][Wy][C], + res (same shape as out, or null) + skip [B][Hs][Ws][C] cropped from the end (or null) */
int noise_ofa_apply(const float* y, const float* coef, int relu6, int Hy, int Wy, const float* res, const float* skip,
                    int Hs, int Ws, float* out, int B, int Ho, int Wo, int C, void* stream);
/* last layer (layers.py:556-564): C -> 1 transposed 3x3, stride 2, padding 1; w = ConvTranspose2d weight [Cmax][1][3][3];
 * out [B][2Hi-1][2Wi-1] raw, slab rows of 2 floats */
int noise_ofa_last_convT(const float* src_y, const float* src_coef, int src_relu6, const float* w, float* out,
                         float* slab, int* rows, int B, int Hi, int Wi, int C, void* stream);
/* network output (layers.py:352-360, :666): one-channel BatchNorm (coef = [scale, shift]) + ReLU6, cropped from the end */
int noise_ofa_final(const float* y, const float* coef, int relu6, float* out, int B, int Hy, int Wy, int H, int W,
                    void* stream);

/* ---- shape-generic forward of configurations outside the tuned schedule: the waveform-domain U-Net of
 * train/configs/mobilenetv2.yaml (train/models.py:1096-1218 under Base, :541-626) runs on the generic convolution
 * kernels above plus the token-level pieces of the transformer bottleneck below -------------------------------------- */
/* Base._normalize (models.py:614-622): mx[b] = max|x[b]| + 1e-6, out = x / mx; x, out [B][L] */
int noise_wave_normalize(const float* x, float* out, float* mx, int B, long L, void* stream);
/* Base.forward's tail (models.py:578-580, :610, :624-626): out [B][H][W] = (y cropped from the end of [B][Hy][Wy] + bias[0])
 * * sigmoid(1e4 * vad[b]) * mx[b]; bias, vad and mx may each be null (term skipped) */
int noise_wave_final(const float* y, const float* bias, const float* vad, const float* mx, float* out, int B, int Hy, int Wy,
                     int H, int W, void* stream);
/* out[t][:D] = LayerNorm(act(y[t] + bias) + res[t]) * gamma + beta (models.py:88-92, :109-118, :214); bias / res may be
 * null, relu != 0 applies max(., 0) before the residual, gamma == beta == null skips the normalisation; pre (optional,
 * [M][D]) receives the LayerNorm's input for the backward pass; D <= 1024 */
int noise_tok_epilogue(const float* y, int ldy, const float* bias, int relu, const float* res, int ldr, const float* gamma,
                       const float* beta, float eps, float* out, int ldo, float* pre, long M, int D, void* stream);
/* ScaledDotProductAttention (models.py:14-36): q, k, v [B*n][ld] with head h in columns [h*dk, (h+1)*dk), dk <= 64, any n;
 * out [B*n][ldo] = softmax(q k^T / temperature) (* keep * keep_scale) v; keep [B][H][n][n] bytes (0 = dropped) or null */
int noise_tok_attention(const float* q, const float* k, const float* v, int ld, const unsigned char* keep, float keep_scale,
                        float* out, int ldo, int B, int n, int H, int dk, float temperature, void* stream);
/* vad[b] = x[b, token 0, :D] . w + bias[0] (models.py:601); x [B][tokens_per_clip][ld] */
int noise_tok_vad(const float* x, int ld, long tokens_per_clip, const float* w, const float* bias, float* vad, int B, int D,
                  void* stream);

/* ---- shape-generic backward: with noise_dw3x3_bwd, noise_first_conv_wgrad and noise_bn_bwd above these differentiate the
 * generic forward, i.e. the waveform-domain training step (models.py:641-685 on the layer stack of :1096-1218).  Gradients
 * of a convolution's raw output come in the lazy form (dz, y_out, coef3) that noise_bn_bwd leaves; coef3 == NULL: dz ---- */
/* out[c] = sum over rows of slab [rows][cols] (the fixed-order fold behind every reduction below) */
int noise_gen_slab_reduce(const float* slab, int rows, long cols, float* out, void* stream);
/* 1x1 layer (models.py:371-381, :395-396): g_in [npix][Cin] = dy [npix][Cout] x Wt^T, Wt = weight transposed [Cin][ldw].
 * Optional epilogue (epi_y != NULL: the layer's input is act(BatchNorm(epi_y)), epi_coef its [scale|shift|mean|invstd|
 * mean_lo]): g_in is masked by that ReLU6 (epi_relu6) and slab gets *rows rows of [sum g | sum g*xhat] (2 Cin floats, at
 * most noise_ofa_max_rows() rows) for noise_bn_bwd_finalize */
int noise_gen_pw_dgrad(const float* dz, const float* y_out, const float* coef3, const float* Wt, int ldw, float* g_in,
                       const float* epi_y, const float* epi_coef, int epi_relu6, float* slab, int* rows, long npix, int Cout,
                       int Cin, void* stream);
/* dW [Cout][Cin] = sum_p dy[p] (x) a[p], a = the layer's lazily normalised input; slab: noise_gen_pw_wgrad_rows() x Cout*Cin */
int noise_gen_pw_wgrad_rows(long npix, int Cout, int Cin);
int noise_gen_pw_wgrad(const float* dz, const float* y_out, const float* coef3, const float* a_y, const float* a_coef,
                       int a_relu6, float* dw, float* slab, long npix, int Cout, int Cin, void* stream);
/* out[c] = sum_m x[m][c] (bias gradients); slab: 256 x N floats */
int noise_gen_colsum(const float* x, int ld, long M, int N, float* out, float* slab, void* stream);
/* out = h > 0 ? g : 0 over n floats (n % 4 == 0): backward of the feed-forward ReLU (models.py:112) */
int noise_gen_relu_bwd(const float* g, const float* h, float* out, long n, void* stream);
/* dst [B][Hd][Wd][C], src [B][Hs][Ws][C]: dst[:, Hd-Hs:, Wd-Ws:, :] += src (gradient of `x + skip[..., -h:, -w:]`, :608) */
int noise_gen_add_cropped(float* dst, const float* src, int B, int Hd, int Wd, int Hs, int Ws, int C, void* stream);
/* LayerNorm backward (models.py:90, :116, :214): out = d/dr of LayerNorm(r) against g (+ add); dgamma [D] followed by dbeta
 * [D] in one buffer; slab: noise_gen_ln_bwd_rows(M) x 2D floats; D <= 1024 */
int noise_gen_ln_bwd_rows(long M);
int noise_gen_ln_bwd(const float* g, const float* r, const float* gamma, float eps, const float* add, float* out, float* dgamma,
                     float* dbeta, float* slab, long M, int D, void* stream);
/* backward of noise_tok_attention (same layouts; o = its output, go = the gradient of o): dq, dk, dv laid out like q, k, v;
 * stats: B*H*n*3 floats of workspace */
int noise_gen_attn_bwd(const float* q, const float* k, const float* v, int ld, const unsigned char* keep, float keep_scale,
                       const float* o, const float* go, int ldo, float* dq, float* dk, float* dv, float* stats, int B, int n, int H,
                       int dk_, float temperature, void* stream);
/* VAD head (models.py:601): gx[b][token 0][:D] += dvad[b] * w, dw[d] = sum_b dvad[b] x[b][token 0][d], db = sum_b dvad[b] */
int noise_gen_vad_bwd(const float* x, float* gx, int ld, long tokens_per_clip, const float* w, const float* dvad, float* dw,
                      float* db, int B, int D, void* stream);
/* last layer (models.py:1164-1171), C -> 1 transposed 3x3 stride 2: dout [B][2Hi-1][2Wi-1] (zero outside the crop), a
 * [B][Hi][Wi][C] plain; g like a, dw (C,1,3,3); slab: noise_gen_last_bwd_rows() x 9C floats */
int noise_gen_last_bwd_rows(void);
int noise_gen_last_convT_bwd(const float* dout, const float* a, const float* w, float* g, float* dw, float* slab, int B, int Hi,
                             int Wi, int C, void* stream);

/* ---- micro-benchmarks of the two rooflines (no counterpart in the reference; not on the product path) ----------- */
/* dst = src (n floats, n % 4 == 0): 8 * n bytes of HBM traffic */
int noise_bench_copy(const float* src, float* dst, long n, void* stream);
/* back-to-back MFMA issue: kind 0 = v_mfma_f32_16x16x4_f32 (2048 flop per wave instruction), kind 1 =
 * v_mfma_f32_16x16x32_f16 (16384 flop), kind 2 = v_mfma_f32_32x32x2_f32 (4096 flop); blocks x 4 waves x iters x 4 instructions; out: blocks * 256 floats */
int noise_bench_mfma(int kind, float* out, int blocks, int iters, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NOISE_HIP_H */
