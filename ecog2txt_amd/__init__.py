"""ecog2txt_amd: MI355X-native backend for the ECoG->text sequence-to-sequence
hot path of jgmakin/ecog2txt (the part the reference delegates to the TF1.x
`machine_learning.SequenceNetwork`, ecog2txt/trainers.py:126-135, 318).

Constants mirror the reference's token/partition contract
(ecog2txt/__init__.py:10-22)."""
import os

# Process-wide defaults that must be in place BEFORE the HIP / HSA runtime initialises (it reads them once): dmabuf IPC (the
# host driver supports nothing else: RCCL's peer-to-peer set-up fails otherwise with `hipIpcGetMemHandle: invalid argument`)
# and the CUs left to RCCL's channel kernels -- the persistent recurrences of the 256-electrode configuration occupy 200
# (forward) / 224 (BPTT) of the 256 CUs for a whole layer sweep, one workgroup each; the exchange gets the other 32.
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('NCCL_MAX_NCHANNELS', '32')

text_dir = os.path.join(os.path.dirname(__file__), 'auxiliary')

EOS_token = '<EOS>'
pad_token = '<pad>'
OOV_token = '<OOV>'

TOKEN_TYPES = {'phoneme', 'word', 'trial', 'word_sequence', 'word_piece_sequence', 'phoneme_sequence'}
DATA_PARTITIONS = {'training', 'validation', 'testing'}
