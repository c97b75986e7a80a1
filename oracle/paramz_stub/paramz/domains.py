_REAL = 'real'
_POSITIVE = 'positive'
_NEGATIVE = 'negative'
_BOUNDED = 'bounded'
