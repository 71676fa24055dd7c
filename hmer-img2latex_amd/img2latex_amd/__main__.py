"""``python -m img2latex_amd predict CHECKPOINT IMAGE ...`` / ``python -m img2latex_amd train ...``."""
import sys

from .cli import main

sys.exit(main())
