"""Writes a small synthetic experiment (manifest + block_breakdowns.json + vocab) into a directory."""
import json
import os

MANIFEST_TEMPLATE = """
{sid}:
  DataGenerator: !!python/name:ecog2txt_amd.data_generators.{generator} ''
  EMA_decay: 0.99
  FF_dropout: 0.1
  N_epochs: {epochs}
  REFERENCE_BIPOLAR: false
  RGB_color: !!python/tuple
  - 0.4
  - 0.65
  - 0.11
  RNN_dropout: 0.3
  TEMPORALLY_CONVOLVE: true
  USE_FIELD_POTENTIALS: false
  USE_LOG_MELS: false
  USE_MFCC_DELTAS: false
  assessment_epoch_interval: {interval}
  bad_electrodes_path: {root}/bad_electrodes_{sid}
  beam_width: 1
  block_types:
    testing: !!set
      mocha-1: null
    training: !!set
      mocha-1: null
      mocha-2: null
    validation: !!set
      mocha-1: null
  data_mapping:
    decoder_targets: text_sequence
    encoder_1_targets: audio_sequence
{extra_map}    encoder_inputs: ecog_sequence
  decimation_factor: null
  encoder_1_targets_penalty_scale: 0.5
{extra_scale}  grid_size:
  - {g0}
  - {g1}
  grid_step: 1
  json_dir: {root}
  layer_sizes:
    decoder_embedding:
    - 16
    decoder_projection: []
    decoder_rnn:
    - 64
    encoder_1_projection:
    - 24
{extra_proj}    encoder_embedding:
{emb}    encoder_rnn:
    - 32
    - 32
  mfcc_winlen: 0.02
  num_cepstral_coeffs: 5
  num_mel_features: 26
  project: SYN
  sampling_rate: {rate}
  sampling_rate_decimated: 16.5
  saved_results_dir: {root}/saved_results
  temperature: 0.384
  text_sequence_vocab_file: vocab.synthetic
  tf_record_partial_path: {root}/tf_records/SYN{sid}_B{{0}}.tfrecord
  tf_summaries_dir: {root}/tf_summaries
  token_type: word_sequence
"""


def make_experiment(root, subject_ids=(401,), epochs=2, interval=1, grid=(4, 4), rate=200, nwords=20, grids=None, extra_aux=False, embedding=(24,),
                    generator='SyntheticSpeechDataGenerator'):
    """grids: {subject id: (rows, cols)} for participants whose electrode grids differ from `grid`; extra_aux: a second
    auxiliary head ('encoder_0_targets', also on the audio sequence, one hidden layer of 12, penalty scale 0.25); embedding:
    layer_sizes['encoder_embedding'] (more than one entry = a stack of strided conv layers); generator: the DataGenerator class of
    ecog2txt_amd.data_generators the manifest names."""
    root = str(root)
    os.makedirs(root, exist_ok=True)
    blocks = {}
    for sid in subject_ids:
        blocks[str(sid)] = {
            '1': {'type': 'mocha-1', 'default_dataset': 'training'},
            '2': {'type': 'mocha-1', 'default_dataset': 'training'},
            '3': {'type': 'mocha-2', 'default_dataset': 'training'},
            '4': {'type': 'mocha-1', 'default_dataset': 'validation'},
            '5': {'type': 'mocha-1', 'default_dataset': 'testing'},
            '6': {'type': 'mocha-3', 'default_dataset': 'training'},      # type not allowed anywhere
            '7': {'type': 'mocha-2', 'default_dataset': 'validation'},    # type not allowed for validation
            '8': {'type': 'mocha-1', 'default_dataset': 'extra'},         # not a partition
        }
        open(os.path.join(root, 'bad_electrodes_%s' % sid), 'w').close()
    with open(os.path.join(root, 'block_breakdowns.json'), 'w') as f:
        json.dump(blocks, f)
    with open(os.path.join(root, 'vocab.synthetic'), 'w') as f:
        f.write('\n'.join(['<pad>', '<EOS>', '<OOV>'] + ['w%03d_' % i for i in range(nwords)]) + '\n')
    path = os.path.join(root, 'experiment.yaml')
    with open(path, 'w') as f:
        for sid in subject_ids:
            g = (grids or {}).get(sid, grid)
            f.write(MANIFEST_TEMPLATE.format(sid=sid, root=root, epochs=epochs, interval=interval, g0=g[0], g1=g[1], rate=rate, generator=generator,
                                            extra_map='    encoder_0_targets: audio_sequence\n' if extra_aux else '',
                                            extra_scale='  encoder_0_targets_penalty_scale: 0.25\n' if extra_aux else '',
                                            extra_proj='    encoder_0_projection:\n    - 12\n' if extra_aux else '',
                                            emb=''.join('    - %d\n' % e for e in embedding)))
    return path
