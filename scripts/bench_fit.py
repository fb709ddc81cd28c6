"""fit()-level throughput at cfg2 sizes: MultiSubjectTrainer.parallel_transfer_learn() on a synthetic participant
(256 electrodes, ~2-s utterances at 200 Hz -> T = 400 after padding, B = 256), i.e. records on disk -> staging ->
HBM-resident partitions -> per-step batch assembly by e2t_gather_rows_u32 -> captured train step.  Prints utterances/s
of the training loop (assessment excluded) next to bench.py's number for the same shapes."""
import os, sys, time, tempfile, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import experiment_fixture as XF
from ecog2txt_amd.data_generators import ECoGDataGenerator, SyntheticSpeechDataGenerator
from ecog2txt_amd.trainers import MultiSubjectTrainer

tmp = tempfile.mkdtemp()
tpl = XF.MANIFEST_TEMPLATE
tpl = re.sub(r'decoder_embedding:\n    - 16', 'decoder_embedding:\n    - 150', tpl)
tpl = re.sub(r'decoder_rnn:\n    - 64', 'decoder_rnn:\n    - 800', tpl)
tpl = re.sub(r'encoder_1_projection:\n    - 24', 'encoder_1_projection:\n    - 225', tpl)
tpl = re.sub(r'encoder_embedding:\n    - 24', 'encoder_embedding:\n    - 100', tpl)
tpl = re.sub(r'encoder_rnn:\n    - 32\n    - 32', 'encoder_rnn:\n    - 400\n    - 400\n    - 400', tpl)
tpl = tpl.replace('num_cepstral_coeffs: 5', 'num_cepstral_coeffs: 13')
XF.MANIFEST_TEMPLATE = tpl
ECoGDataGenerator.text_dir = tmp
SyntheticSpeechDataGenerator.num_sentences = 50
SyntheticSpeechDataGenerator.trials_per_block = 352            # 3 training blocks -> 1056 utterances
SyntheticSpeechDataGenerator.min_seconds, SyntheticSpeechDataGenerator.max_seconds_synth = 1.5, 2.0
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
path = XF.make_experiment(tmp, subject_ids=(401,), epochs=epochs, interval=10 ** 6, grid=(16, 16), nwords=1803)
ck = os.path.join(tmp, 'ck'); os.makedirs(ck)
# BENCH_FIT_OPTIONS="k=v,k=v": engine options for an A/B (e.g. prefetch_batches=False)
eopts = dict(kv.split('=', 1) for kv in os.environ.get('BENCH_FIT_OPTIONS', '').split(',') if kv)
tr = MultiSubjectTrainer(path, [401], checkpoint_dir=ck, VERBOSE=False, SN_kwargs={'N_cases': 256, 'max_hyp_length': 10, 'engine_options': eopts},
                         DG_kwargs={'max_samples': 400})
t0 = time.perf_counter()
for s in tr.ecog_subjects:
    s.write_tf_records_maybe()
t_write = time.perf_counter() - t0
net = tr.net
orig_assess = net._assess
t_assess = [0.0]
def timed_assess(*a, **k):
    torch.cuda.synchronize(); t = time.perf_counter()
    r = orig_assess(*a, **k)
    torch.cuda.synchronize(); t_assess[0] += time.perf_counter() - t
    return r
net._assess = timed_assess
orig_stage = net._stage
t_stage = [0.0]
def timed_stage(*a, **k):
    t = time.perf_counter(); r = orig_stage(*a, **k); t_stage[0] += time.perf_counter() - t
    return r
net._stage = timed_stage
steps = [0]
eng_step = None
def run(n_epochs):
    net.N_epochs = n_epochs
    torch.cuda.synchronize(); t = time.perf_counter()
    a = tr.parallel_transfer_learn()
    torch.cuda.synchronize()
    return time.perf_counter() - t, a
wall1, _ = run(1)                          # warm-up: staging, graph capture
a0, s0 = t_assess[0], t_stage[0]
t_assess[0] = t_stage[0] = 0.0
# the training loop is timed directly: from the first train_step of the fit to the end of the fit (device idle), minus the
# assessments in between -- not by subtracting the staging time from the wall time (staging and the first steps overlap on the
# host side since the partitions are resident: the subtraction under-counted the loop by 20 % in round 3's first measurement)
eng = net._engine
orig_step = eng.train_step
t_first, n_calls, a_at_first = [None], [0], [0.0]
def timed_step(*a, **k):
    if t_first[0] is None:
        torch.cuda.synchronize(); t_first[0] = time.perf_counter(); a_at_first[0] = t_assess[0]
    n_calls[0] += 1
    return orig_step(*a, **k)
eng.train_step = timed_step
# ceiling probes (wrong training, timing only): BENCH_FIT_HACK=nogather -- the batch assembly skipped after the first epoch;
# noloss -- the per-epoch loss read-back (a device -> host sync) answered from a cached value; both
hack = os.environ.get('BENCH_FIT_HACK', '')
if 'nogather' in hack or hack == 'both':
    orig_load, n_load = net._load_batch, [0]
    def load_once(*a, **k):
        n_load[0] += 1
        if n_load[0] <= 8:
            return orig_load(*a, **k)
    net._load_batch = load_once
if 'noloss' in hack or hack == 'both':
    orig_losses, cached = eng.losses, [None]
    def losses_cached(ws):
        if cached[0] is None:
            cached[0] = orig_losses(ws)
        return cached[0]
    eng.losses = losses_cached
orig_save, t_save = net._save, [0.0]
def timed_save(*a, **k):
    torch.cuda.synchronize(); t = time.perf_counter(); r = orig_save(*a, **k); t_save[0] += time.perf_counter() - t
    return r
net._save = timed_save
wall, res = run(epochs)
t_end = time.perf_counter()
eng.train_step = orig_step
loop_s = (t_end - t_first[0]) - (t_assess[0] - a_at_first[0]) - t_save[0]          # (the checkpoint at the end of fit is not the loop)
d = net._stage(tr.ecog_subjects[-1], 'training')
n_train = d['n']
steps_per_epoch = -(-n_train // 256)
train_s = loop_s
assert n_calls[0] == epochs * steps_per_epoch, (n_calls[0], epochs, steps_per_epoch)
print('records written in %.1f s; first fit (staging + capture + 1 epoch) %.1f s' % (t_write, wall1))
print('training partition: %d utterances, T=%d, C=%d, L=%d; %d steps/epoch of B=256' % (n_train, d['T'], d['X'].shape[2], d['L'], steps_per_epoch))
print('%d epochs: wall %.3f s, of which staging (records -> padded arrays) %.3f s, assessment %.3f s, checkpoint %.3f s, training loop (first step .. last step done) %.3f s' % (epochs, wall, t_stage[0], t_assess[0], t_save[0], train_s))
print('fit-level: %.3f ms per step, %.0f utterances/s over the training loop (padding utterances of the last batch counted as work: %.0f real utterances/s)'
      % (1e3 * train_s / (epochs * steps_per_epoch), epochs * steps_per_epoch * 256 / train_s, epochs * n_train / train_s))
print('losses first/last epoch:', res['training'].losses[0], res['training'].losses[-1])
