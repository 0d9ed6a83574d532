"""CLI mirror of the reference's examples/inference.py (flags --task/-t --input/-i --ref-audio/-ra --ref-text/-rt
--video/-v --output/-o --model/-m --device/-d --no-reuse; exit code 0/1, reference examples/inference.py:152-235)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_MODEL = None


def inference(task, input_text, model_path, ref_audio=None, ref_text=None, video=None, output="./output", device=0, reuse=True):
    global _MODEL
    from unimoe_audio_amd.api import UniMoEAudio
    try:
        if _MODEL is None or not reuse:
            _MODEL = UniMoEAudio(model_path, device)
        if task == "text_to_speech":
            return _MODEL.text_to_speech(input_text, ref_text, ref_audio, output_dir=output)
        if task == "text_to_music":
            return _MODEL.text_to_music(input_text, output_dir=output)
        if task == "video_text_to_music":
            return _MODEL.video_text_to_music(video, input_text, output_dir=output)
        raise ValueError(f"unknown task {task}")
    except Exception as e:   # the reference swallows every exception and returns None (examples/inference.py:116-118)
        print(f"inference failed: {e}")
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", "-t", required=True, choices=["text_to_speech", "text_to_music", "video_text_to_music"])
    ap.add_argument("--input", "-i", required=True)
    ap.add_argument("--ref-audio", "-ra")
    ap.add_argument("--ref-text", "-rt")
    ap.add_argument("--video", "-v")
    ap.add_argument("--output", "-o", default="./output")
    ap.add_argument("--model", "-m", required=True)
    ap.add_argument("--device", "-d", type=int, default=0)
    ap.add_argument("--no-reuse", action="store_true")
    a = ap.parse_args()
    out = inference(a.task, a.input, a.model, a.ref_audio, a.ref_text, a.video, a.output, a.device, not a.no_reuse)
    sys.exit(0 if out else 1)


if __name__ == "__main__":
    main()
