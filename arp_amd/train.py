"""Path (2): return-conditioned policy train step (mirror of create_train_step,
/root/reference/arp_dt/main_procgen.py:104-141).  Under construction."""


def smoke():
    raise NotImplementedError
