"""ecog2txt_amd: MI355X-native backend for the ECoG->text sequence-to-sequence
hot path of jgmakin/ecog2txt (the part the reference delegates to the TF1.x
`machine_learning.SequenceNetwork`, ecog2txt/trainers.py:126-135, 318).

Constants mirror the reference's token/partition contract
(ecog2txt/__init__.py:10-22)."""
import os

# The one process-wide default set at import (documented in README.md): it must be in place BEFORE the HIP / HSA runtime
# initialises, which reads it once -- dmabuf IPC (the host driver supports nothing else: RCCL's peer-to-peer set-up fails
# otherwise with `hipIpcGetMemHandle: invalid argument`).  (NCCL_MAX_NCHANNELS is defaulted where a communicator is made:
# parallel.RcclSync.)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

text_dir = os.path.join(os.path.dirname(__file__), 'auxiliary')

EOS_token = '<EOS>'
pad_token = '<pad>'
OOV_token = '<OOV>'

TOKEN_TYPES = {'phoneme', 'word', 'trial', 'word_sequence', 'word_piece_sequence', 'phoneme_sequence'}
DATA_PARTITIONS = {'training', 'validation', 'testing'}
