import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffusiontexturepainting_amd import _lib
torch.zeros(1, device="cuda")
lib = ctypes.CDLL(_lib.LIB_PATH)
for ku in (1, 2):
    for geglu in (0, 1):
        print("LN  K =", 320 * ku, "geglu" if geglu else "plain", "->", lib.dtp_debug_lnlin_occupancy(ku, geglu, 1), "workgroups per CU")
    print("raw K =", 320 * ku, "->", lib.dtp_debug_lnlin_occupancy(ku, 0, 0), "workgroups per CU")
