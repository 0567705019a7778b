"""Prompt strings of the reference tasks (data, not code): /root/reference/arp_dt/data_procgen.py:281-317."""

_CLIP_INSTRUCT = {
    "coinrun": "the goal is to collect the coin.",
    "coinrun_aisc": "the goal is to collect the coin.",
    "maze": "navigate a maze to collect the yellow cheese.",
    "maze_aisc": "navigate a maze to collect the yellow cheese.",
    "maze_yellowline": "navigate a maze to collect the yellow line.",
    "maze_redline_yellowgem": "navigate a maze to collect the red line.",
}

_IS_COINRUN = lambda env: "coinrun" in env  # noqa: E731  (substring test, as data_procgen.py:303,309,312,315)

# (inst_type, env predicate or None for "any env") -> prompt; first match wins.  The strings are the reference's byte for byte
# (typos included: they are what the shipped reward datasets were labelled with).
_SPECIAL_INSTRUCT = (
    ("random1", None, "His voice echoed through the empty hallway."),
    ("random2", None, "NeurIPS 2023 will be held again at the at the New Orleans Ernest N. Morial Convention Center."),
    ("misinfo", _IS_COINRUN, "The agent must go to the far right of the level."),
    ("misinfo", "maze_aisc".__eq__, "navigate a maze to reacth to the top right corner."),
    ("misinfo", "maze_yellowline".__eq__, "navigate a maze to collect yellow gem."),
    ("misinfo2", _IS_COINRUN, "The goal is to collect the red strawberry."),
    ("misinfo3", _IS_COINRUN, "The goal is to reach the saw."),
    ("misinfo4", _IS_COINRUN, "The goal is to jump as high as you can."),
)


def get_clip_instruct(task):
    """data_procgen.py:281-293 (returns None for an unknown task, like the reference's if-chain)."""
    return _CLIP_INSTRUCT.get(task)


def get_clip_special_instruct(env_name, inst_type):
    """data_procgen.py:296-317: the prompt of the first matching (inst_type, env) row; no row -> the reference's ValueError."""
    for kind, env_ok, prompt in _SPECIAL_INSTRUCT:
        if kind == inst_type and (env_ok is None or env_ok(env_name)):
            return prompt
    raise ValueError("You must pass any condition.")
