import sys

from .serving.cli import main

sys.exit(main())
