#!/usr/bin/env python3
"""`python main.py ...` — same command line as the reference's contrastive_video_textures/main.py."""
import avtex  # noqa: F401
from avtex.main import cli

if __name__ == "__main__":
    cli()
