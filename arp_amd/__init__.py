"""arp_amd -- MI355X-native (gfx950) hot paths of ARP-DT behind a C ABI (libarp_hip.so).

Path (1): CLIP reward labelling (``arp_amd.label_reward``, ``arp_amd.clip``).
Path (2): return-conditioned policy train step (``arp_amd.train``).
Importing the compute modules requires the built HIP extension; there is no CPU fallback.
"""
__version__ = "0.1.0"
