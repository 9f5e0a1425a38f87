"""ORACLE — test infrastructure only.  CPU restatement of the reference's hot
path (head forward, matchers, criterion, LSAP).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product package ``svol_amd`` never does."""
