import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print(cols)
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
msk = [(s, e) for n, s, e in rows if 'msk_demod' in n]
fir = [(s, e) for n, s, e in rows if 'fir_u8' in n]
msk = msk[-40:]
gaps = [msk[i+1][0] - msk[i][1] for i in range(len(msk)-1)]
durs = [e - s for s, e in msk]
print('msk launches', len(msk), 'avg dur us', sum(durs)/len(durs)/1e3)
print('gaps us:', [round(g/1e3,1) for g in gaps])
# FIR placement relative to MSK
fir = fir[-40:]
for i in range(len(msk)-8, len(msk)):
    s, e = msk[i]
    ov = [(fs - s, fe - s) for fs, fe in fir if fe > s and fs < e]
    print('msk', i, 'dur', round((e-s)/1e3,1), 'fir overlapping (start,end rel us):', [(round(a/1e3,1), round(b/1e3,1)) for a, b in ov])
