"""Per-subject parameter objects handed to the network backend (TF-free).

Mirrors the attribute surface the reference's backend reads from each subject
(ecog2txt/subjects.py:56-62 and SURVEY.md Appendix C): `subnet_id`, `block_ids`,
`decimation_factor`, `tf_record_partial_path`, `data_manifests` (+ `data_generator`,
`block_types`, `pretrain_all_blocks`).  Counters, sub-grid masks and plotting helpers of the
reference are outside the hot path and are not restated."""
import json
import os

import numpy as np

from . import EOS_token, pad_token, OOV_token, DATA_PARTITIONS
from .toolbox import auto_attribute, str2int_hook

CONTINUOUS_TYPES = ('ecog_sequence', 'audio_sequence')


def string_seq_to_index_seq(seq, feature_list, append_ids, oov_id):
    """bytes/str tokens -> indices into feature_list, unknown -> oov_id, then append_ids
    (role of tfh.string_seq_to_index_seq at reference subjects.py:355-361)."""
    lookup = {t: i for i, t in enumerate(feature_list)}
    out = [lookup.get(t.decode('utf-8') if isinstance(t, bytes) else t, oov_id) for t in seq]
    return np.array(out + list(append_ids), dtype=np.int64)


class SequenceDataManifest:
    """How one sequence type is unpacked from a record, and a few derived sizes
    (reference subjects.py:274-404)."""

    @auto_attribute
    def __init__(self, sequence_type, num_features=None, num_features_raw=None, transform=None, padding_value=None,
                 penalty_scale=1.0, distribution=None, mask=None, get_feature_list=None, APPEND_EOS=False):
        pass

    # -- feature kind as stored in the record (subjects.py:297-302) --
    @property
    def is_continuous(self):
        return self.sequence_type in CONTINUOUS_TYPES

    @property
    def feature_value(self):
        return 'float32' if self.is_continuous else 'bytes'

    # -- widths (subjects.py:304-336) --
    @property
    def num_features(self):
        if self.mask is not None:
            return len(self.mask.inds)
        if self.get_feature_list is not None:
            return len(self.get_feature_list())
        return self._num_features

    @num_features.setter
    def num_features(self, n):
        self._num_features = n

    @property
    def num_features_raw(self):
        if self._num_features_raw is not None:
            return self._num_features_raw
        if self.mask is not None:
            return self._num_features
        if self.get_feature_list is not None:
            return 1                      # one class index per step
        return self.num_features

    @num_features_raw.setter
    def num_features_raw(self, n):
        self._num_features_raw = n

    # -- record value -> network value (subjects.py:338-367) --
    @property
    def transform(self):
        if self._transform is not None:
            return self._transform
        if self.mask is not None:
            inds = np.asarray(self.mask.inds)
            return lambda seq: np.asarray(seq)[:, inds]
        if self.get_feature_list is not None:
            feats = list(self.get_feature_list())
            oov = feats.index(OOV_token) if OOV_token in feats else 2          # subjects.py:348-351
            tail = [feats.index(EOS_token)] if self.APPEND_EOS else []
            return lambda seq: string_seq_to_index_seq(seq, feats, tail, oov)
        return lambda seq: seq

    @transform.setter
    def transform(self, fn):
        self._transform = fn

    # -- output distribution (subjects.py:369-384) --
    @property
    def distribution(self):
        if self._distribution is not None:
            return self._distribution
        return {'ecog_sequence': 'Rayleigh', 'audio_sequence': 'Gaussian'}.get(self.sequence_type, 'categorical')

    @distribution.setter
    def distribution(self, d):
        self._distribution = d

    # -- padding (subjects.py:386-404) --
    @property
    def padding_value(self):
        if self._padding_value is not None:
            return self._padding_value
        if self.get_feature_list is None:
            return 0.0
        feats = list(self.get_feature_list())
        return feats.index(pad_token) if pad_token in feats else 0

    @padding_value.setter
    def padding_value(self, v):
        self._padding_value = v


class ECoGSubject:
    """Attributes for one participant + its data generator (reference subjects.py:27-181)."""

    @auto_attribute(CHECK_MANIFEST=True)
    def __init__(self, manifest, subj_id, pretrain_all_blocks=False, input_mask=None, target_specs=(), block_ids=(),
                 block_types=None, data_mapping=None, decimation_factor=None, sampling_rate_decimated=None,
                 json_dir=None, _DG_kwargs=()):
        with open(os.path.join(self.json_dir, 'block_breakdowns.json')) as f:
            self._block_dict = json.load(f, object_hook=str2int_hook)[subj_id]
        generator_class = manifest['DataGenerator']
        self.data_generator = generator_class(manifest, subj_id, **dict(_DG_kwargs))
        self.target_specs = dict(target_specs)
        self.data_manifests = {
            key: (SequenceDataManifest(**spec) if isinstance(spec, dict) else SequenceDataManifest(spec))
            for key, spec in self.data_mapping.items()}

    # ---- read by the backend ----
    @property
    def subnet_id(self):
        return self.subj_id

    @property
    def block_ids(self):
        """{partition: set(block)}: a block belongs to a partition iff that is its default_dataset AND its
        type is allowed for the partition; all-but-last subjects train on every selected block
        (reference subjects.py:110-138; trainers.py:76)."""
        if self._block_ids:
            return self._block_ids
        sel = {part: {blk for blk, info in self._block_dict.items()
                      if info['default_dataset'] == part and info['type'] in self.block_types[part]}
               for part in DATA_PARTITIONS}
        if self.pretrain_all_blocks:
            sel['training'] = set().union(*sel.values())
        if self.target_specs:
            every = set().union(*sel.values())
            sel = {part: every for part in DATA_PARTITIONS}
        return sel

    @block_ids.setter
    def block_ids(self, v):
        self._block_ids = v

    @property
    def tf_record_partial_path(self):
        return self.data_generator.tf_record_partial_path

    @property
    def decimation_factor(self):
        """explicit value wins; else round(sampling_rate / sampling_rate_decimated) (subjects.py:144-153)."""
        if self._decimation_factor is not None:
            return self._decimation_factor
        return int(np.round(self.data_generator.sampling_rate / self.sampling_rate_decimated))

    @decimation_factor.setter
    def decimation_factor(self, v):
        self._decimation_factor = v

    @property
    def data_manifests(self):
        """ECoG / audio widths follow the data generator on every access (subjects.py:159-177)."""
        for dm in self._data_manifests.values():
            if dm.sequence_type == 'ecog_sequence':
                dm.num_features = self.data_generator.num_ECoG_channels
            elif dm.sequence_type == 'audio_sequence':
                dm.num_features = self.data_generator.num_MFCC_features
        return self._data_manifests

    @data_manifests.setter
    def data_manifests(self, v):
        self._data_manifests = v

    @property
    def input_mask(self):
        return self._input_mask

    @input_mask.setter
    def input_mask(self, m):
        self._input_mask = m
        if m is not None:
            m.good_channels = self.data_generator.good_channels

    # ---- record writing (subjects.py:183-196) ----
    def write_tf_records_maybe(self, sequence_type=None, data_partitions=DATA_PARTITIONS):
        if sequence_type is None:
            sequence_type = self.data_manifests['decoder_targets'].sequence_type
        class_list = []
        for part in data_partitions:
            class_list = self.data_generator.write_to_Protobuf_maybe(sequence_type, self.block_ids[part])
        return class_list
