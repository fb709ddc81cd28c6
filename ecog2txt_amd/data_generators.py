"""TF-free shell data generator + a synthetic subclass.

`ECoGDataGenerator` keeps the attribute surface that feeds shapes and paths into the hot path
(reference ecog2txt/data_generators.py:45-245: electrode layout, good channels, bipolar map,
max_samples, MFCC width, vocab lookup, record path) and the record writer; the three
user-supplied hooks (`_ecog_token_generator`, `_get_wav_data`, `_query`, reference :502-530)
stay abstract.  MFCC / log-mel extraction and word-piece encoding are restated in
speech_features.py and reached through `_get_MFCC_features` / `_sentence_tokenize` / `TokenEncoder`."""
import os

import zlib

import numpy as np

from . import text_dir as _default_text_dir
from .toolbox import auto_attribute
from . import tfrecord

MAX_SECONDS = {'phoneme': 0.2, 'word': 1.0, 'word_sequence': 6.25, 'word_piece_sequence': 6.25,
               'phoneme_sequence': 6.25, 'trial': 6.25}            # data_generators.py:35-42


class ECoGDataGenerator:
    text_dir = None          # override to relocate vocab files (defaults to the package's auxiliary dir)

    @auto_attribute(CHECK_MANIFEST=True)
    def __init__(self, manifest, subj_id, grid_step=None, num_cepstral_coeffs=None, mfcc_winlen=None, USE_LOG_MELS=None,
                 USE_MFCC_DELTAS=None, USE_FIELD_POTENTIALS=None, REFERENCE_BIPOLAR=None, num_mel_features=None,
                 sampling_rate=None, token_type=None, bad_electrodes_path=None, tf_record_partial_path=None,
                 grid_size=None, max_seconds=None, max_samples=None, good_electrodes=None):
        for key, value in manifest.items():                         # '<sequence_type>_vocab_file' keys
            if key.endswith('_vocab_file'):
                setattr(self, key, value)

    # ---- geometry (data_generators.py:103-109, 173-233, 490-500) ----
    @property
    def elec_layout(self):
        n = int(np.prod(self.grid_size))
        step = self.grid_step or 1
        return np.arange(n - 1, -1, -1).reshape(self.grid_size).T[::step, ::step]

    @property
    def bipolar_to_elec_map(self):
        lay = self.elec_layout
        pairs = []
        for i in range(lay.shape[0]):
            for j in range(lay.shape[1]):
                if j + 1 < lay.shape[1]:
                    pairs.append((lay[i, j], lay[i, j + 1]))
                if i + 1 < lay.shape[0]:
                    pairs.append((lay[i, j], lay[i + 1, j]))
        return np.array(pairs)

    @property
    def bad_electrodes_path(self):
        return self._bad_electrodes_path

    @bad_electrodes_path.setter
    def bad_electrodes_path(self, p):
        self._bad_electrodes_path = p

    @property
    def good_electrodes(self):
        """0-indexed set; the bad-electrode file is 1-indexed (data_generators.py:173-193)."""
        if self._good_electrodes is not None:
            return self._good_electrodes
        bad = set()
        if self.bad_electrodes_path and os.path.isfile(self.bad_electrodes_path):
            with open(self.bad_electrodes_path) as f:
                bad = {int(line) - 1 for line in f if line.strip()}
        return set(range(int(np.prod(self.grid_size)))) - bad

    @good_electrodes.setter
    def good_electrodes(self, s):
        self._good_electrodes = s

    @property
    def good_channels(self):
        good = self.good_electrodes
        order = self.elec_layout.flatten().tolist()
        if self.USE_FIELD_POTENTIALS:
            M = len(order)
            kept = [e for e in order if e in good]
            return kept + [e + M for e in kept]
        if self.REFERENCE_BIPOLAR:
            return [ch for ch, pair in enumerate(self.bipolar_to_elec_map) if all(e in good for e in pair)]
        return [e for e in order if e in good]

    @property
    def num_ECoG_channels(self):
        return len(self.good_channels)

    # ---- lengths and feature widths (data_generators.py:139-171) ----
    @property
    def max_seconds(self):
        return self._max_seconds if self._max_seconds is not None else MAX_SECONDS.get(self.token_type, 0.2)

    @max_seconds.setter
    def max_seconds(self, v):
        self._max_seconds = v

    @property
    def max_samples(self):
        if self._max_samples is not None:
            return self._max_samples
        return int(np.floor(self.sampling_rate * self.max_seconds))

    @max_samples.setter
    def max_samples(self, v):
        self._max_samples = v

    @property
    def num_MFCC_features(self):
        if self.USE_LOG_MELS:
            return self.num_mel_features + 1
        return 2 * self.num_cepstral_coeffs if self.USE_MFCC_DELTAS else self.num_cepstral_coeffs

    @property
    def target_type(self):
        return 'Trial' if 'sequence' in self.token_type else self.token_type.capitalize()

    # ---- paths (data_generators.py:122-137, 235-245) ----
    @property
    def tf_record_partial_path(self):
        p = self._tf_record_partial_path
        if self.REFERENCE_BIPOLAR and (self.grid_step or 1) > 1:
            return os.path.join(os.path.dirname(p), 'lowdensity_bipolar', os.path.basename(p))
        return p

    @tf_record_partial_path.setter
    def tf_record_partial_path(self, p):
        self._tf_record_partial_path = p

    def sequence_type_to_vocab_file_path(self, sequence_type):
        name = getattr(self, sequence_type + '_vocab_file', None)
        if name is None:
            return None
        path = os.path.join(self.text_dir or _default_text_dir, name)
        return path if os.path.isfile(path) else None

    def get_class_list(self, sequence_type=None, block_set=None):
        if sequence_type is not None:
            path = self.sequence_type_to_vocab_file_path(sequence_type)
            if self.token_type == 'word_piece_sequence':          # data_generators.py:431-433
                return self.TokenEncoder(path)._all_subtoken_strings
            with open(path) as f:
                return f.read().split()
        if block_set is not None:
            return self.write_to_Protobuf_maybe(sequence_type, block_set)
        raise ValueError('get_class_list needs a sequence_type or a block_set')

    def TokenEncoder(self, vocab_file_path):
        """tensor2tensor's SubwordTextEncoder, restated (data_generators.py:475-485)."""
        from .speech_features import SubwordTextEncoder
        return SubwordTextEncoder(vocab_file_path)

    def _get_MFCC_features(self, index, winstep, nfft=512):
        """MFCCs (or log-mels + log energy, optionally with deltas) of trial `index`'s audio at frame step `winstep`
        seconds: data_generators.py:328-380, parameters as fixed there; `_get_wav_data` is the subclass hook."""
        from .speech_features import mfcc_features
        audio_sampling_rate, audio_signal = self._get_wav_data(index)
        if audio_signal is None:
            return np.zeros((0, self.num_MFCC_features))
        if self.num_MFCC_features == 0:
            return np.zeros((int(audio_signal.shape[0] / audio_sampling_rate / winstep), 0))
        return mfcc_features(audio_signal, audio_sampling_rate, self.mfcc_winlen, winstep, self.num_mel_features,
                             self.num_cepstral_coeffs, bool(self.USE_LOG_MELS), bool(self.USE_MFCC_DELTAS), nfft)

    def _sentence_tokenize(self, token_list, sequence_type=None):
        """lower-cased word + '_' as UTF-8 bytes; word pieces through the subword encoder (data_generators.py:446-473)."""
        if self.token_type == 'word_piece_sequence':
            enc = self.TokenEncoder(self.sequence_type_to_vocab_file_path(sequence_type))
            pieces = enc._all_subtoken_strings
            return [pieces[i].encode('utf-8') for i in enc.encode(' '.join(t.lower() for t in token_list))]
        if self.token_type == 'trial':
            return [' '.join(t.lower() + '_' for t in token_list).encode('utf-8')]
        return [(t.lower() + '_').encode('utf-8') for t in token_list]

    # ---- padded tensors and records (data_generators.py:247-326, 382-425) ----
    def get(self, block_set, sequence_types=None):
        sequence_types = sequence_types or ['ecog_sequence']
        n = self._query(block_set)
        widths = {'ecog_sequence': self.num_ECoG_channels, 'audio_sequence': self.num_MFCC_features}
        out = {st: (np.zeros((n, self.max_samples, widths[st])) if st in widths else []) for st in sequence_types}
        i = 0
        for block in block_set:
            for element in self._ecog_token_generator(block):
                for st, store in out.items():
                    assert st in element, 'sequence_type %s is not produced by the generator' % st
                    tok = element[st]
                    if isinstance(store, list):
                        store.append(tok)
                    else:
                        store[i, :tok.shape[0]] = tok[:self.max_samples]
                i += 1
        return out

    def _write_to_Protobuf(self, block):
        path = self.tf_record_partial_path.format(block)
        os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
        # written under a private name and moved into place: the ranks of a data-parallel run may all find the file missing
        # at once (their generators are deterministic, so the contents agree) and must never read a half-written one
        tmp = '%s.tmp.%d' % (path, os.getpid())
        with tfrecord.TFRecordWriter(tmp) as w:
            for example in self._ecog_token_generator(block):
                w.write(tfrecord.encode_example(example))
        os.replace(tmp, path)

    def write_to_Protobuf_maybe(self, sequence_type, block_set):
        """Write missing block files, then return the unique tokens of `sequence_type` in them."""
        seen = set()
        for block in block_set:
            path = self.tf_record_partial_path.format(block)
            if not os.path.exists(path):
                self._write_to_Protobuf(block)
            for payload in tfrecord.tf_record_iterator(path):
                for tok in tfrecord.decode_example(payload).get(sequence_type, []):
                    seen.add(tok.decode('utf-8') if isinstance(tok, bytes) else str(tok))
        return list(seen)

    # ---- hooks a subclass provides (data_generators.py:502-530) ----
    def _get_wav_data(self, index):
        return None, None

    def _query(self, block_set):
        return None

    def _ecog_token_generator(self, block):
        return iter(())


class SyntheticSpeechDataGenerator(ECoGDataGenerator):
    """Deterministic synthetic participant: a fixed set of sentences from the
    vocabulary; ECoG = |N(0,1)| 'high-gamma' plus a sentence-dependent low-rank signal so that the
    mapping is learnable; 13-dim pseudo-MFCC targets at the input rate."""
    num_sentences = 50
    trials_per_block = 40
    min_words, max_words = 3, 9
    min_seconds, max_seconds_synth = 1.2, 2.0
    vocab_words = None        # optional explicit word list (without specials)

    def _rng(self, *key):
        # (zlib.crc32, not hash(): PYTHONHASHSEED randomises str hashes per process, and every rank must see the same participant)
        return np.random.default_rng([zlib.crc32(str(k).encode()) for k in (self.subj_id,) + key])

    def _sentences(self):
        vocab = self.vocab_words
        if vocab is None:
            path = self.sequence_type_to_vocab_file_path('text_sequence')
            if path:
                with open(path) as f:
                    vocab = [w.rstrip('_') for w in f.read().split() if not w.startswith('<')]
            else:
                vocab = ['w%03d' % i for i in range(200)]
        rng = np.random.default_rng(12345)          # same sentences for every subject
        return [[vocab[j] for j in rng.integers(0, len(vocab), size=rng.integers(self.min_words, self.max_words + 1))]
                for _ in range(self.num_sentences)]

    def _query(self, block_set):
        return self.trials_per_block * len(list(block_set))

    def _ecog_token_generator(self, block):
        for _, example in self._trials(block):
            yield example

    def _trials(self, block):
        """(sentence index, example dict) of every trial of a block."""
        sents = self._sentences()
        C, K = self.num_ECoG_channels, max(self.num_MFCC_features, 0)
        basis = self._rng('basis').standard_normal((len(sents), 3, C))
        rng = self._rng('block', block)
        for _ in range(self.trials_per_block):
            si = int(rng.integers(0, len(sents)))
            T = int(rng.uniform(self.min_seconds, self.max_seconds_synth) * self.sampling_rate)
            T = min(T, self.max_samples)
            tt = np.linspace(0, 1, T)[:, None]
            x = np.abs(rng.standard_normal((T, C)))
            x = (x - 0.8) / 0.6 + 0.8 * (np.sin(2 * np.pi * (1 + si % 4) * tt) * basis[si, 0] + tt * basis[si, 1] + basis[si, 2])
            x[np.abs(x).max(1) == 0] = 1e-3           # a genuine all-zero row would read as padding
            example = {'ecog_sequence': x.astype(np.float32), 'text_sequence': self._sentence_tokenize(sents[si])}
            if K:
                a = np.cumsum(rng.standard_normal((T, K)), 0) / np.sqrt(np.arange(1, T + 1))[:, None] + basis[si, 0, :K] * tt
                a[np.abs(a).max(1) == 0] = 1e-3
                example['audio_sequence'] = a.astype(np.float32)
            yield si, example


class SyntheticWaveformDataGenerator(SyntheticSpeechDataGenerator):
    """The synthetic participant with REAL acoustic features as auxiliary targets: every trial carries a synthetic speech-like
    waveform (sentence-dependent formant-like sinusoids under a syllable-rate envelope, plus noise; 16 kHz) through the
    `_get_wav_data` hook, and its `audio_sequence` is `_get_MFCC_features(index, 1 / sampling_rate)` -- the reference's
    feature path (ecog2txt/data_generators.py:328-380: one frame per ECoG sample, `mfcc_winlen`, `num_mel_features`,
    `num_cepstral_coeffs`, `USE_LOG_MELS`, `USE_MFCC_DELTAS` from the manifest), restated in speech_features.py."""
    audio_sampling_rate = 16000

    def _trial_plan(self, block):
        """(sentence index, number of ECoG samples) of every trial of a block, in the order _ecog_token_generator yields them."""
        if not hasattr(self, '_plans'):
            self._plans = {}
        if block not in self._plans:
            self._plans[block] = list(self._trials(block))
        return self._plans[block]

    def _get_wav_data(self, index):
        """index = (block, trial number).  The waveform lasts as long as the trial's ECoG."""
        block, trial = index
        si, ex = self._trial_plan(block)[trial]
        T = ex['ecog_sequence'].shape[0]
        n = int(round(T / self.sampling_rate * self.audio_sampling_rate))
        t = np.arange(n) / self.audio_sampling_rate
        rng = self._rng('wave', block, trial)
        f0 = 110.0 + 7.0 * (si % 9)
        formants = [300.0 + 90.0 * (si % 5), 1200.0 + 160.0 * (si % 7), 2500.0 + 110.0 * (si % 3)]
        env = 0.55 + 0.45 * np.sin(2 * np.pi * (2.5 + 0.3 * (si % 4)) * t) ** 2          # syllable-rate envelope
        sig = sum(a * np.sin(2 * np.pi * f * t + 0.3 * np.sin(2 * np.pi * f0 * t)) for a, f in zip((1.0, 0.6, 0.3), formants))
        sig = env * sig * (1.0 + 0.2 * np.sign(np.sin(2 * np.pi * f0 * t))) + 0.05 * rng.standard_normal(n)
        return self.audio_sampling_rate, (3000.0 * sig).astype(np.float64)

    def _ecog_token_generator(self, block):
        for trial, (_, ex) in enumerate(self._trial_plan(block)):
            out = dict(ex)
            if self.num_MFCC_features:
                a = self._get_MFCC_features((block, trial), 1.0 / self.sampling_rate)
                T = ex['ecog_sequence'].shape[0]
                feat = np.zeros((T, self.num_MFCC_features))
                feat[:min(T, a.shape[0])] = a[:T]
                if a.shape[0] and a.shape[0] < T:
                    feat[a.shape[0]:] = a[-1]                    # (python_speech_features yields window/step - 1 fewer frames than ECoG samples: hold the last one)
                feat[np.abs(feat).max(1) == 0] = 1e-3           # a genuine all-zero row would read as padding
                out['audio_sequence'] = feat.astype(np.float32)
            yield out
