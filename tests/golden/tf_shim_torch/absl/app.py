"""absl.app stand-in."""
import sys


def run(main):
    return main(sys.argv)
