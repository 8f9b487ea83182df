#!/usr/bin/env python3
"""debug: do engines leak file descriptors / memory / threads?  create, use, close 80 engines in one process"""
import os, sys, resource
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import zkvm_pairings_amd as z
from zkvm_pairings_amd import synthetic
def stat():
    st = open("/proc/self/status").read()
    g = lambda k: [l.split()[1] for l in st.splitlines() if l.startswith(k)][0]
    return "fds %d  threads %s  VmRSS %s kB  VmSize %s kB  maps %d" % (len(os.listdir("/proc/self/fd")), g("Threads"), g("VmRSS"), g("VmSize"), sum(1 for _ in open("/proc/self/maps")))
print("limits: nofile", resource.getrlimit(resource.RLIMIT_NOFILE), "memlock", resource.getrlimit(resource.RLIMIT_MEMLOCK), "nproc", resource.getrlimit(resource.RLIMIT_NPROC))
print("start:", stat())
g1 = g2 = None
for i in range(80):
    e = z.PairingEngine(0, kernel="coop")
    if g1 is None:
        g1, g2, _, _ = synthetic.random_pairs(e, 300, seed=1)
    e.pairing(g1, g2)
    if i % 3 == 0:
        a = e.host_array((300, 72)); e.pairing(g1, g2, out=a); del a
    e.close()
    if i % 10 == 9:
        print("after %d engines:" % (i + 1), stat())
