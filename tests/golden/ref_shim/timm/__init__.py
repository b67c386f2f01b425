__version__ = '0.9.2-standin'
