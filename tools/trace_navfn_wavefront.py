import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
out=[]; started=False; n=0
for r in rows:
    name=r['Kernel_Name']
    if 'k_navfn_wf_init' in name:
        started=True; out=[]; t0=int(r['Start_Timestamp'])
    if started:
        out.append((name.split('(')[0][-22:], (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
        if 'k_navfn_wf_path' in name:
            n+=1
            if n==2:
                rd=[o for o in out if 'wf_round' in o[0]]
                print("rounds", len(rd), "sum %.1f us" % sum(o[2] for o in rd), "active mean %.1f" % (sum(o[2] for o in rd if o[2]>8)/max(1,len([o for o in rd if o[2]>8]))))
                print("path %.1f us; span init->path end %.1f us" % (out[-1][2], out[-1][1]+out[-1][2]))
                for o in out[:6]+out[-4:]: print("%-24s start %8.1f us  dur %7.1f us" % o)
                break
            started=False
