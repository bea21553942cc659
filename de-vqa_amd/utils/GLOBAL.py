"""Paths (R/utils/GLOBAL.py:1-6).  ROOT_PATH is this package; model paths come from the
environment (DEVQA_BLIP2_PATH ...) because the reference hard-codes absolute author paths."""
import os

ROOT_PATH = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
model_path_map = {
    "blip2-opt-2.7b": os.environ.get("DEVQA_BLIP2_PATH", "models/blip2-opt-2.7b-hg"),
    "llava-v1.5-7b": os.environ.get("DEVQA_LLAVA_PATH", "models/llava-v1.5-7b-hf"),
    "minigpt-4-vicuna-7b": os.environ.get("DEVQA_MINIGPT4_PATH", "models/MiniGPT-4"),
}
