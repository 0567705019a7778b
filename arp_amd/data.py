"""Prompt strings of the reference tasks (data, not code): /root/reference/arp_dt/data_procgen.py:281-317."""

_CLIP_INSTRUCT = {
    "coinrun": "the goal is to collect the coin.",
    "coinrun_aisc": "the goal is to collect the coin.",
    "maze": "navigate a maze to collect the yellow cheese.",
    "maze_aisc": "navigate a maze to collect the yellow cheese.",
    "maze_yellowline": "navigate a maze to collect the yellow line.",
    "maze_redline_yellowgem": "navigate a maze to collect the red line.",
}


def get_clip_instruct(task):
    """data_procgen.py:281-293 (returns None for an unknown task, like the reference's if-chain)."""
    return _CLIP_INSTRUCT.get(task)


def get_clip_special_instruct(env_name, inst_type):
    """data_procgen.py:296-300."""
    if inst_type == "random1":
        return "His voice echoed through the empty hallway."
    elif inst_type == "random2":
        return "NeurIPS 2023 will be held again at the at the New Orleans Ernest N. Morial Convention Center."
