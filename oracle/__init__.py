# TEST INFRASTRUCTURE: CPU oracle (see fem_oracle.py header).  Never imported
# by flow_amd/.
