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
    """data_procgen.py:296-317, including the ValueError when no branch returns."""
    if inst_type == "random1":
        return "His voice echoed through the empty hallway."
    elif inst_type == "random2":
        return "NeurIPS 2023 will be held again at the at the New Orleans Ernest N. Morial Convention Center."
    elif inst_type == "misinfo":
        if "coinrun" in env_name:
            return "The agent must go to the far right of the level."
        elif env_name == "maze_aisc":
            return "navigate a maze to reacth to the top right corner."
        elif env_name == "maze_yellowline":
            return "navigate a maze to collect yellow gem."
    elif inst_type == "misinfo2":
        if "coinrun" in env_name:
            return "The goal is to collect the red strawberry."
    elif inst_type == "misinfo3":
        if "coinrun" in env_name:
            return "The goal is to reach the saw."
    elif inst_type == "misinfo4":
        if "coinrun" in env_name:
            return "The goal is to jump as high as you can."
    raise ValueError("You must pass any condition.")
