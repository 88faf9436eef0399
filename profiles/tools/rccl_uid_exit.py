"""Does a process that called ncclGetUniqueId (ptnn_comm_unique_id) and never ncclCommInitRank still exit?  rccl_uid_exit.py [init]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptnn_amd import _lib
t0 = time.time()
uid = _lib.comm_unique_id()
print(f"unique id after {time.time()-t0:.1f} s; stage: {_lib.comm_last_stage()}", flush=True)
print("leaving", flush=True)
