"""Diagnostics scripts run on the DEBUG build of the library (csrc/build.sh with E2T_DEBUG=1 -> libecog2txt_hip_dbg.so): the
kernel-variant switches (E2T_GEMM_*, E2T_CONV_FWD, E2T_BIG_SPREAD_*) and the phase-stamp buffer of the recurrences
(E2T_LSTM_DBG) do not exist in the product library.  Import this module BEFORE ecog2txt_amd."""
import os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
subprocess.run(['bash', os.path.join(ROOT, 'ecog2txt_amd', 'csrc', 'build.sh')], check=True, env=dict(os.environ, E2T_DEBUG='1'),
               stdout=subprocess.DEVNULL)
os.environ['E2T_DEBUG_LIB'] = '1'
