"""Acoustic target features and word-piece tokens, TF-free and without the reference's third-party helpers
(offline CPU preprocessing that produces the `audio_sequence` auxiliary targets and the
`word_piece_sequence` decoder targets).

Reference: `ECoGDataGenerator._get_MFCC_features` (ecog2txt/data_generators.py:328-380) calls
`python_speech_features.fbank / lifter / delta` and `scipy.fftpack.dct` with parameters fixed in its code
(preemph 0.97, rectangular window, nfft 512, lowfreq 0, highfreq None = Nyquist, ceplifter 22, DCT-II 'ortho', c0
replaced by the log frame energy, optional deltas with N = 2, or log-mels + log energy); `_sentence_tokenize`
(data_generators.py:446-485) encodes `word_piece_sequence` targets with tensor2tensor's `SubwordTextEncoder`.
Neither package is vendored, pinned or installed here, so their PUBLISHED algorithms are restated
below ([RECALL]: python_speech_features 0.6 `base.py` / `sigproc.py`; tensor2tensor 1.x
`data_generators/text_encoder.py` / `tokenizer.py`); tests/test_speech_features.py pins the pieces against closed forms
and against scipy's DCT.
"""
import math
import unicodedata

import numpy as np


# ---------------------------------------------------------------------------------------------------------------------
# python_speech_features, restated
# ---------------------------------------------------------------------------------------------------------------------
def _round_half_up(x):
    return int(math.floor(x + 0.5))


def hz2mel(hz):
    return 2595.0 * np.log10(1.0 + np.asarray(hz, dtype=np.float64) / 700.0)


def mel2hz(mel):
    return 700.0 * (10.0 ** (np.asarray(mel, dtype=np.float64) / 2595.0) - 1.0)


def preemphasis(signal, coeff=0.97):
    signal = np.asarray(signal, dtype=np.float64)
    return np.append(signal[0], signal[1:] - coeff * signal[:-1])


def framesig(signal, frame_len, frame_step, winfunc=lambda n: np.ones((n,))):
    """Overlapping frames [numframes, frame_len]; the tail is zero-padded to a whole frame."""
    slen = len(signal)
    frame_len, frame_step = _round_half_up(frame_len), _round_half_up(frame_step)
    numframes = 1 if slen <= frame_len else 1 + int(math.ceil((1.0 * slen - frame_len) / frame_step))
    padlen = int((numframes - 1) * frame_step + frame_len)
    padded = np.concatenate((np.asarray(signal, dtype=np.float64), np.zeros(padlen - slen)))
    idx = np.arange(frame_len)[None, :] + (np.arange(numframes) * frame_step)[:, None]
    return padded[idx] * winfunc(frame_len)[None, :]


def powspec(frames, nfft):
    """1/nfft * |rfft|^2 per frame (frames longer than nfft are truncated by the transform, as numpy does)."""
    return (1.0 / nfft) * np.square(np.abs(np.fft.rfft(frames, nfft)))


def get_filterbanks(nfilt=26, nfft=512, samplerate=16000, lowfreq=0, highfreq=None):
    """Triangular mel filters [nfilt, nfft//2 + 1] with corner bins floor((nfft+1) * f / samplerate)."""
    highfreq = highfreq or samplerate / 2
    melpoints = np.linspace(hz2mel(lowfreq), hz2mel(highfreq), nfilt + 2)
    bins = np.floor((nfft + 1) * mel2hz(melpoints) / samplerate)
    fb = np.zeros((nfilt, nfft // 2 + 1))
    for j in range(nfilt):
        for i in range(int(bins[j]), int(bins[j + 1])):
            fb[j, i] = (i - bins[j]) / (bins[j + 1] - bins[j])
        for i in range(int(bins[j + 1]), int(bins[j + 2])):
            fb[j, i] = (bins[j + 2] - i) / (bins[j + 2] - bins[j + 1])
    return fb


def fbank(signal, samplerate=16000, winlen=0.025, winstep=0.01, nfilt=26, nfft=512, lowfreq=0, highfreq=None,
          preemph=0.97, winfunc=lambda n: np.ones((n,))):
    """(mel filterbank energies [frames, nfilt], total frame energies [frames]); exact zeros become machine epsilon."""
    highfreq = highfreq or samplerate / 2
    frames = framesig(preemphasis(signal, preemph), winlen * samplerate, winstep * samplerate, winfunc)
    pspec = powspec(frames, nfft)
    energy = pspec.sum(1)
    energy = np.where(energy == 0, np.finfo(float).eps, energy)
    feat = pspec @ get_filterbanks(nfilt, nfft, samplerate, lowfreq, highfreq).T
    return np.where(feat == 0, np.finfo(float).eps, feat), energy


def dct2_ortho(x):
    """DCT-II along axis 1 with the orthonormal scaling (scipy.fftpack.dct(x, type=2, axis=1, norm='ortho'))."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[1]
    k = np.arange(n)
    basis = np.cos(np.pi * (2 * k[None, :] + 1) * k[:, None] / (2.0 * n))       # [coefficient, sample]
    scale = np.full(n, math.sqrt(2.0 / n))
    scale[0] = math.sqrt(1.0 / n)
    return x @ (basis * scale[:, None]).T


def lifter(cepstra, L=22):
    if L <= 0:
        return cepstra
    n = np.arange(cepstra.shape[1])
    return (1.0 + (L / 2.0) * np.sin(np.pi * n / L)) * cepstra


def delta(feat, N):
    """Regression deltas over +-N frames with edge padding."""
    if N < 1:
        raise ValueError('N must be an integer >= 1')
    denom = 2 * sum(i * i for i in range(1, N + 1))
    padded = np.pad(feat, ((N, N), (0, 0)), mode='edge')
    w = np.arange(-N, N + 1, dtype=np.float64)
    return np.stack([w @ padded[t:t + 2 * N + 1] for t in range(feat.shape[0])], 0) / denom


def mfcc_features(audio_signal, audio_sampling_rate, winlen, winstep, num_mel_features=26, num_cepstral_coeffs=13,
                  USE_LOG_MELS=False, USE_MFCC_DELTAS=False, nfft=512):
    """The feature matrix of ecog2txt/data_generators.py:354-378 for one utterance's audio."""
    features, energy = fbank(audio_signal, audio_sampling_rate, winlen, winstep, num_mel_features, nfft, 0, None, 0.97,
                             lambda n: np.ones((n,)))
    features = np.log(features)
    if not USE_LOG_MELS:
        features = dct2_ortho(features)[:, :num_cepstral_coeffs]
        features = lifter(features, 22)
        features[:, 0] = np.log(energy)
    else:
        features = np.concatenate((features, np.log(energy)[:, None]), axis=1)
    return np.concatenate((features, delta(features, N=2)), axis=1) if USE_MFCC_DELTAS else features


# ---------------------------------------------------------------------------------------------------------------------
# tensor2tensor SubwordTextEncoder (encoding side), restated
# ---------------------------------------------------------------------------------------------------------------------
def _is_alnum(c):
    return unicodedata.category(c)[0] in ('L', 'N')


def tokenizer_encode(text):
    """Split where alphanumeric-ness changes; a single space between two alphanumeric runs is dropped."""
    if not text:
        return []
    ret, start = [], 0
    flags = [_is_alnum(c) for c in text]
    for pos in range(1, len(text)):
        if flags[pos] != flags[pos - 1]:
            token = text[start:pos]
            if token != ' ' or start == 0:
                ret.append(token)
            start = pos
    ret.append(text[start:])
    return ret


_ESCAPE_CHARS = set('\\_u;0123456789')


class SubwordTextEncoder:
    """Greedy longest-match word-piece encoder over a vocabulary file of one (optionally quoted) subtoken per line, as
    tensor2tensor writes it; `_all_subtoken_strings` is the class list the reference reads (data_generators.py:432-433)."""

    def __init__(self, vocab_file_path=None, subtoken_strings=None):
        if vocab_file_path is not None:
            subtoken_strings = []
            with open(vocab_file_path, encoding='utf-8') as f:
                for line in f:
                    s = line.rstrip('\n').strip()
                    if len(s) >= 2 and ((s[0] == "'" and s[-1] == "'") or (s[0] == '"' and s[-1] == '"')):
                        s = s[1:-1]
                    subtoken_strings.append(s)
        self._all_subtoken_strings = list(subtoken_strings)
        self._subtoken_string_to_id = {s: i for i, s in enumerate(self._all_subtoken_strings) if s}
        self._max_subtoken_len = max((len(s) for s in self._all_subtoken_strings), default=0)
        self._alphabet = {c for s in self._all_subtoken_strings for c in s} | _ESCAPE_CHARS

    @property
    def vocab_size(self):
        return len(self._all_subtoken_strings)

    def _escape_token(self, token):
        token = token.replace('\\', '\\\\').replace('_', '\\u')
        return ''.join(c if (c in self._alphabet and c != '\n') else '\\%d;' % ord(c) for c in token) + '_'

    def _escaped_token_to_subtoken_ids(self, escaped):
        ret, start, n = [], 0, len(escaped)
        while start < n:
            for end in range(min(n, start + self._max_subtoken_len), start, -1):
                sid = self._subtoken_string_to_id.get(escaped[start:end])
                if sid is not None:
                    ret.append(sid)
                    start = end
                    break
            else:
                raise ValueError('token substring %r not found in the subtoken vocabulary' % escaped[start:])
        return ret

    def encode(self, text):
        ids = []
        for token in tokenizer_encode(text):
            ids += self._escaped_token_to_subtoken_ids(self._escape_token(token))
        return ids
