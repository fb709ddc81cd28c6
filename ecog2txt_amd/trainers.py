"""`MultiSubjectTrainer`: orchestration of (transfer) learning across participants, restated over
the MI355X backend.  Same constructor, kwargs routing and public methods as the reference
orchestrator (ecog2txt/trainers.py:41-442, 556-602, 703-732, 757-859, 925-949); the TensorFlow-only
pieces (graph rebuilding, TF checkpoint reader, SavedModel loading) are replaced by reads of this backend's
device buffers and checkpoints.

Usage is the README's (reference README.md:72-102):

    trainer = MultiSubjectTrainer('my_manifest.yaml', subject_ids=[400, 401],
                                  SN_kwargs={'FF_dropout': 0.4}, DG_kwargs={...}, ES_kwargs={...})
    for subject in trainer.ecog_subjects:
        subject.write_tf_records_maybe()
    assessments = trainer.parallel_transfer_learn()
"""
import os
import pickle
import re
from collections import defaultdict
from functools import partial, reduce

import numpy as np

from . import text_dir, TOKEN_TYPES, DATA_PARTITIONS, EOS_token, pad_token, OOV_token
from . import tfrecord
from .manifests import load_manifest
from .subjects import ECoGSubject
from .sequence_network import SequenceNetwork, target_inds_to_sequences, EMA_SUFFIX  # noqa: F401 (re-export)


def _identity(x):
    return x


class MultiSubjectTrainer:
    def __init__(self, experiment_manifest_name, subject_ids, checkpoint_dir='.', restore_epoch=None, SN_kwargs=(),
                 DG_kwargs=(), RP_kwargs=(), ES_kwargs=(), VERBOSE=True, **kwargs):
        SN_kwargs = dict(SN_kwargs)
        self.experiment_manifest = load_manifest(experiment_manifest_name, text_dir)       # trainers.py:60-61
        token_type = self.experiment_manifest[subject_ids[-1]]['token_type']
        assert token_type in TOKEN_TYPES, 'Unrecognized token_type!!'
        self._token_type = token_type
        self._RP_kwargs = dict(RP_kwargs)

        # every subject but the last pretrains on all of its blocks (trainers.py:72-82)
        self.ecog_subjects = [
            ECoGSubject(self.experiment_manifest[sid], sid, pretrain_all_blocks=(sid != subject_ids[-1]),
                        **dict(ES_kwargs), _DG_kwargs=dict(DG_kwargs))
            for sid in subject_ids]

        self.VERBOSE = VERBOSE
        self.checkpoint_dir = checkpoint_dir
        self.restore_epoch = restore_epoch

        # experiment-specific adjustments of the data manifests (trainers.py:94-103)
        for subject in self.ecog_subjects:
            for data_key, dm in subject.data_manifests.items():
                if data_key == 'decoder_targets' and 'sequence' in token_type:
                    dm.APPEND_EOS = True
                scale = self.experiment_manifest[subject.subnet_id].get(data_key + '_penalty_scale')
                if scale is not None:
                    dm.penalty_scale = scale
        self.set_feature_lists(**kwargs)

        # the net is built from the LAST subject's manifest (trainers.py:126-135)
        self.net = SequenceNetwork(
            self.experiment_manifest[subject_ids[-1]], EOS_token=EOS_token, pad_token=pad_token, OOV_token=OOV_token,
            training_GPUs=[int(os.environ.get('LOCAL_RANK', 0))], TARGETS_ARE_SEQUENCES='sequence' in token_type,
            VERBOSE=VERBOSE, **SN_kwargs)
        self.checkpoint_dir = checkpoint_dir        # again, to set the net's checkpoint_path
        self._results_plotter = None

    def vprint(self, *args, **kwargs):
        if self.VERBOSE:
            print(*args, **kwargs)

    # ------------------------------------------------------------------ vocabularies (trainers.py:147-211)
    def set_feature_lists(self, **kwargs):
        for subject in self.ecog_subjects:
            for data_key, dm in subject.data_manifests.items():
                if dm.distribution != 'categorical':
                    continue
                st = dm.sequence_type
                kw_name = st + '_vocab_list'
                file_path = subject.data_generator.sequence_type_to_vocab_file_path(st)
                pkl_path = os.path.join(self.checkpoint_dir, st + '_vocab_file.pkl')
                if kw_name in kwargs:                                   # 1: explicit list
                    source, class_list = 'argument ' + kw_name, kwargs[kw_name]
                elif file_path is not None:                             # 2: vocab file in text_dir
                    source, class_list = file_path, subject.data_generator.get_class_list(st)
                elif os.path.isfile(pkl_path):                          # 3: pickled list next to the checkpoints
                    with open(pkl_path, 'rb') as fp:
                        class_list = [t.decode('utf-8') for t in pickle.load(fp)]
                    source = pkl_path
                else:                                                   # 4: from the records themselves
                    specials = ([pad_token, EOS_token, OOV_token]
                                if 'sequence' in self._token_type and 'encoder_' not in data_key else [pad_token, OOV_token])
                    class_list = self._training_intersection_validation_union(st, specials)
                    source = 'training-intersection/validation-union'
                self.vprint('Setting feature_list for %s to %s' % (data_key, source))
                dm.get_feature_list = partial(_identity, class_list)    # picklable (trainers.py:203-207)

    def _training_intersection_validation_union(self, sequence_type, special_tokens=()):
        per_partition = []
        for part in DATA_PARTITIONS:
            sets = [set(s.write_tf_records_maybe(sequence_type, [part])) for s in self.ecog_subjects]
            per_partition.append(reduce((lambda a, b: a & b) if part == 'training' else (lambda a, b: a | b), sets))
        tokens = sorted(t for t in set().union(*per_partition) if t not in special_tokens)
        return list(special_tokens) + tokens

    # ------------------------------------------------------------------ checkpoint bookkeeping (trainers.py:213-256)
    @property
    def checkpoint_dir(self):
        net = getattr(self, 'net', None)
        if net is not None:
            net.checkpoint_path = os.path.join(self._checkpoint_dir, 'model.ckpt')
        return self._checkpoint_dir

    @checkpoint_dir.setter
    def checkpoint_dir(self, d):
        self._checkpoint_dir = d
        self.checkpoint_dir

    @property
    def restore_epoch(self):
        if self._restore_epoch is not None:
            return self._restore_epoch
        epochs = sorted(int(n.split('-')[1].split('.')[0]) for n in os.listdir(self.checkpoint_dir)
                        if n.split('-')[0] == 'model.ckpt' and n.split('.')[-1] == 'index')
        return epochs[-1] if epochs else None

    @restore_epoch.setter
    def restore_epoch(self, e):
        self._restore_epoch = e

    # ------------------------------------------------------------------ learning schedules
    def parallel_transfer_learn(self, RESUME=False, fit_kwargs=()):
        """All subjects in one fit (multi-task); trainers.py:303-327."""
        if RESUME:
            fit_kwargs = {'_restore_epoch': self.restore_epoch, **dict(fit_kwargs),
                          'train_vars_scope': 'seq2seq', 'reuse_vars_scope': 'seq2seq'}
            self.ecog_subjects = [self.ecog_subjects[-1]]
        assessments = self.net.fit(self.ecog_subjects, **dict(fit_kwargs))
        self._save_results(assessments)
        if self._restore_epoch is not None:
            self.restore_epoch = self.restore_epoch + self.net.N_epochs if RESUME else self.net.N_epochs
        return assessments

    def sequential_transfer_learn(self, pretraining_epochs=60, training_epochs=200, posttraining_epochs=340):
        """One subject after the other; a new subject first trains only its own sub-network with the shared
        body restored (regex scopes as in trainers.py:329-374)."""
        proprietary, reusable = 'seq2seq/subnet', 'seq2seq/(?!subnet)'
        fit_kwargs, latest, assessments = {}, 0, None
        for subject in self.ecog_subjects:
            if subject is self.ecog_subjects[0]:
                fit_kwargs['reuse_vars_scope'] = None
            else:
                self.net.N_epochs = pretraining_epochs
                fit_kwargs.update(train_vars_scope=proprietary, reuse_vars_scope=reusable, _restore_epoch=latest)
                self.net.fit([subject], **fit_kwargs)
                latest += self.net.N_epochs
                fit_kwargs.update(_restore_epoch=latest, reuse_vars_scope='seq2seq')
            if subject is self.ecog_subjects[-1]:
                training_epochs += posttraining_epochs
            self.net.N_epochs = training_epochs
            fit_kwargs['train_vars_scope'] = 'seq2seq'
            assessments = self.net.fit([subject], **fit_kwargs)
            latest += self.net.N_epochs
            self._save_results(assessments)
        self.restore_epoch = latest
        return assessments

    def assess_saved_model(self):
        self.update_net_from_saved_model()
        return self.net.restore_and_assess(self.ecog_subjects, self.restore_epoch)

    # ------------------------------------------------------------------ sizes from a checkpoint (trainers.py:383-554)
    def update_net_from_saved_model(self):
        self.net.layer_sizes, data_sizes, strides, EMA = self.recover_model_sizes()
        self.net.TEMPORALLY_CONVOLVE = len(next(iter(strides.values()), []))
        if self.net.TEMPORALLY_CONVOLVE > 1:                  # a conv stack: the layers' strides are the widths of their rank-4 weights
            self.net.encoder_strides = [int(x) for x in next(iter(strides.values()))]
        self.net.EMA_decay = 0.99 * EMA
        for subject in self.ecog_subjects:
            sid = str(subject.subnet_id)
            for key, size in data_sizes.get(sid, {}).items():
                if key in subject.data_manifests:
                    subject.data_manifests[key].num_features = size
            for key, size in data_sizes.get(None, {}).items():
                if key in subject.data_manifests:
                    subject.data_manifests[key].num_features = size
            if strides.get(sid):
                subject.decimation_factor = int(np.prod(strides[sid]))

    def recover_model_sizes(self):
        """Walk the checkpoint's variable->shape map with the reference's naming grammar: outer scope seq2seq,
        optional subnet_<id>, '<subnet>_<in>_<out>_<layer>/weights', RNN variables under cell_<k> with 4 packed
        gates, transposed last projection layer, rank-4 conv weights whose width is the stride."""
        prefix = '%s-%d' % (os.path.join(self.checkpoint_dir, 'model.ckpt'), self.restore_epoch)
        if os.path.exists(prefix + '.npz'):
            z = np.load(prefix + '.npz')
            shapes = {name: z[name].shape for name in z.files}
        else:       # a TensorFlow checkpoint (e.g. written by the reference): the index alone has names and shapes
            from . import tf_checkpoint
            shapes = dict(tf_checkpoint.list_variables(prefix))
        info = defaultdict(lambda: defaultdict(dict))
        EMA = False
        for name, shape in shapes.items():
            if name.startswith('__'):
                continue
            scopes = name.split('/')
            if scopes[-1] == 'ExponentialMovingAverage':
                EMA = True
                continue
            if scopes.pop(0) != 'seq2seq':
                continue
            sub = scopes.pop(0)
            sid = None
            if re.match(r'subnet_', sub):
                sid, sub = sub.split('_', 1)[1], scopes.pop(0)
            cell = next((sc for sc in scopes if re.match(r'cell_\d+', sc)), None)
            if cell is not None:
                if scopes[-1] != 'kernel':
                    continue
                layer = int(cell.split('_')[1])
            elif scopes[0] == 'weights':
                sub, _, _, layer = sub.rsplit('_', 3)
                layer = int(layer)
            else:
                continue
            if cell is not None and len(scopes) > 2:
                sub = sub + '/' + scopes[0]                       # fw / bw directions share a layer size
            info[sid][sub][layer] = shape
        layer_sizes, data_sizes, strides = {}, defaultdict(dict), defaultdict(list)
        for sid, subnets in info.items():
            for sub, layers in subnets.items():
                base = sub.split('/')[0]
                if sub.endswith('/bw'):
                    continue
                sizes = []
                for layer in sorted(layers):
                    shp = layers[layer]
                    if '_projection' in base and layer == max(layers):
                        data_sizes[sid][base.replace('_projection', '_targets')] = shp[0]
                        continue
                    sizes.append(shp[-1] // 4 if '_rnn' in base else shp[-1])
                    if base == 'encoder_embedding':
                        if len(shp) == 4:
                            strides[sid].append(shp[1])
                        if layer == min(layers):
                            data_sizes[sid]['encoder_inputs'] = shp[-2]
                layer_sizes[base] = sizes
        enc = [layer_sizes.pop(k)[0] for k in sorted(k for k in layer_sizes if re.fullmatch(r'encoder_rnn_\d+', k))]
        layer_sizes['encoder_rnn'] = enc
        return layer_sizes, data_sizes, strides, EMA

    # ------------------------------------------------------------------ results table (trainers.py:556-602)
    def _save_results(self, assessments):
        subject = self.ecog_subjects[-1]
        manifest = self.experiment_manifest[subject.subnet_id]
        save_dir = manifest['saved_results_dir']
        os.makedirs(save_dir, exist_ok=True)
        name = '_'.join(['accuracies', manifest['project'] + '-'.join(str(s.subnet_id) for s in self.ecog_subjects),
                         str(self.net.FF_dropout), str(self.net.RNN_dropout)]
                        + ['-'.join(str(n) for n in sizes) for _, sizes in sorted(self.net.layer_sizes.items())])
        path = os.path.join(save_dir, name)
        self.vprint('save file is ' + path)
        interval = self.net.assessment_epoch_interval
        n = len(assessments['training'].decoder_accuracies)
        np.savetxt(path, np.stack([assessments['training'].decoder_accuracies,
                                   assessments['training'].decoder_word_error_rates,
                                   assessments['validation'].decoder_accuracies,
                                   assessments['validation'].decoder_word_error_rates,
                                   np.arange(0, n * interval, interval)], axis=1), fmt='%.4f',
                   header='training accs | training WERs | validation acc | validation WERs | epochs')
        return path

    # ------------------------------------------------------------------ probes
    def get_saliencies(self, contrib_method, assessment_type='norms'):
        """Average input-electrode saliency for one output (trainers.py:703-732): all target penalties are
        zeroed except the one named by `contrib_method` ('..._saliency_map' -> '..._targets')."""
        subject = self.ecog_subjects[-1]
        old = {}
        for key, dm in subject.data_manifests.items():
            if '_targets' in key:
                old[key] = dm.penalty_scale
                dm.penalty_scale = 0.0
        subject.data_manifests[contrib_method.replace('saliency_map', 'targets')].penalty_scale = 1.0
        try:
            return self.net.restore_and_get_saliencies([subject], self.restore_epoch, data_partition='validation',
                                                       assessment_type=assessment_type)
        finally:
            for key, v in old.items():
                subject.data_manifests[key].penalty_scale = v

    def get_internal_activations(self):
        """'convolved_inputs', 'reversed_inputs', 'decimated_reversed_targets', 'final_RNN_state' on the last
        subject's validation blocks with the saved model (trainers.py:757-859; the reference rebuilds these graph
        snippets in TF, here they are read back from the device buffers of a forward pass)."""
        self.update_net_from_saved_model()
        return self.net.restore_and_get_internal_activations(self.ecog_subjects, self.restore_epoch,
                                                              data_partition='validation')

    def construct_online_predictor(self, targets_list=None, subject=None, max_length=None):
        """predict(inputs) -> hypotheses for raw ECoG [T,C] / [B,T,C] with the saved model (the reference builds this
        from a SavedModel export, trainers.py:925-949; here from this trainer's checkpoint).  targets_list defaults
        to the decoder vocabulary, so the hypotheses are word strings."""
        self.update_net_from_saved_model()
        subject = subject or self.ecog_subjects[-1]
        if targets_list is None:
            targets_list = list(subject.data_manifests['decoder_targets'].get_feature_list())
        return self.net.online_predictor(self.ecog_subjects, self.restore_epoch, targets_list=targets_list,
                                         subject=subject, max_length=max_length)

    def subject_to_table(self):
        import pandas as pd
        rows = []
        for s in self.ecog_subjects:
            d = {key: dm.num_features for key, dm in s.data_manifests.items()}
            d.update({dm.sequence_type + '_vocab_list': dm.get_feature_list() for dm in s.data_manifests.values()
                      if dm.distribution == 'categorical'})
            d.update(block_types=s.block_types, block_ids=s.block_ids, decimation_factor=s.decimation_factor,
                     restore_epoch=self.restore_epoch)
            rows.append(pd.Series(d, name=s.subnet_id))
        return pd.concat(rows, axis=1).transpose()

    def tf_record_to_numpy_data(self, subj_id, block_id):
        """Iterate one block's records as numpy dicts (floats reshaped, strings left as bytes; trainers.py:861-922)."""
        subject = next((s for s in self.ecog_subjects if s.subj_id == subj_id), None)
        if subject is None:
            raise ValueError('Requested subject not in this trainer')
        for payload in tfrecord.tf_record_iterator(subject.tf_record_partial_path.format(block_id)):
            raw = tfrecord.decode_example(payload)
            out = {}
            for key, dm in subject.data_manifests.items():
                v = raw[dm.sequence_type]
                out[key] = np.asarray(v, np.float32).reshape(-1, dm.num_features_raw) if dm.is_continuous else v
            yield out
