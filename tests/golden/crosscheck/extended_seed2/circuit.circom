
template Program() {
   signal input CuSaSOOy;
   signal input OGWiyKqf;
   signal input urDgbMov;
   signal input mdJMrqPu;
   for (var cawxPgik = 21; (cawxPgik<=51); cawxPgik=(cawxPgik+1)) {
      var tcRgYCbE = (3801462687429696038*3460086721918963735);
      var RClsbTWA = (15978962999394941129-11048676178656897600);
      RClsbTWA=(11925592123400272513-4017085629454261397);
      for (var fsBwYDpm = 11; (fsBwYDpm<=41); fsBwYDpm=(fsBwYDpm+1)) {
         for (var iNOpAuZl = 29; (iNOpAuZl<=49); iNOpAuZl=(iNOpAuZl+1)) {
            fsBwYDpm=(5260112035621119752*17695553914810211714);
         }
         RClsbTWA=(17863132340689033660+4898398629125743456);
      }
   }
   var HyWavpLw[10];
   var UteodODK[1];
   signal output LvKIEUqx;
   LvKIEUqx<==(((15939026911459456457+OGWiyKqf)*(OGWiyKqf+16600124587323430179))+(mdJMrqPu-urDgbMov));
   UteodODK[0]=(5334209129747575313-11705413272793949487);
   signal output GheDjtER;
   GheDjtER<==(((mdJMrqPu-CuSaSOOy)*LvKIEUqx)+(mdJMrqPu-mdJMrqPu));
   for (var RrSAhxrX = 23; (RrSAhxrX<=56); RrSAhxrX=(RrSAhxrX+1)) {
      RrSAhxrX=(16886160926421420287+12787176457473729479);
      for (var fOKKgGcP = 12; (fOKKgGcP<=46); fOKKgGcP=(fOKKgGcP+1)) {
         signal output zBnObiZb;
         zBnObiZb<==((OGWiyKqf*(LvKIEUqx-2905753820633757983))+(CuSaSOOy+mdJMrqPu));
         UteodODK[0]=(UteodODK[0]);
         fOKKgGcP=(UteodODK[0]);
      }
   }
   for (var mDnSXuhx = 11; (mDnSXuhx<=41); mDnSXuhx=(mDnSXuhx+1)) {
      UteodODK[0]=(17956112542284095286-17246175008089274406);
   }
   for (var XcrjxDYp = 26; (XcrjxDYp<=35); XcrjxDYp=(XcrjxDYp+1)) {
      XcrjxDYp=GheDjtER;
      UteodODK[0]=(3523788033551196011*16513754949990396758);
      signal output KoJPpAno;
      KoJPpAno<==((16471258040983990514*urDgbMov)+LvKIEUqx);
      XcrjxDYp=(UteodODK[0]);
   }
   UteodODK[0]=13872662537718701002;
   HyWavpLw[3]=5049320214125608898;
   for (var tjzBrwEO = 28; (tjzBrwEO<=60); tjzBrwEO=(tjzBrwEO+1)) {
      signal output hWADsMln;
      hWADsMln<==((urDgbMov*(GheDjtER+urDgbMov))+(CuSaSOOy-mdJMrqPu));
   }
   for (var iLXQXcrR = 5; (iLXQXcrR<=15); iLXQXcrR=(iLXQXcrR+1)) {
      HyWavpLw[9]=11640843699201119707;
      for (var uJzXrCNo = 20; (uJzXrCNo<=27); uJzXrCNo=(uJzXrCNo+1)) {
         signal output yERwXhKF;
         yERwXhKF<==((OGWiyKqf*mdJMrqPu)+(15564047833313047337+GheDjtER));
         signal output HBnkWCRv;
         HBnkWCRv<==((CuSaSOOy*(5666179561070890594-15573254321701496134))+(13786508783349117605+LvKIEUqx));
         signal output SBnKYroS;
         SBnKYroS<==(((mdJMrqPu+OGWiyKqf)*(12015100743625581652+10176210666538746861))+(mdJMrqPu+15698071700594383984));
         for (var DOoFbIgC = 4; (DOoFbIgC<=6); DOoFbIgC=(DOoFbIgC+1)) {
            var RmIzgHWP = (11213947731683742726*4694792032156426017);
            var FPmsigxx[5];
            FPmsigxx[1]=7971261303139066293;
         }
      }
   }
   var psvZUTsE[9];
   var HpgRrQzb[2];
   signal output syZNAVkJ;
   syZNAVkJ<==(((15161376133630110362-CuSaSOOy)*GheDjtER)+(4013172491533740410+OGWiyKqf));
   HyWavpLw[3]=(11718876945611298192-5629113178508678687);
   signal output cjfbetpo;
   cjfbetpo<==((6436554167883173339*CuSaSOOy)+syZNAVkJ);
   signal output IaBRrIab;
   IaBRrIab<==((8587394379082902974*(11528269922447072510-OGWiyKqf))+CuSaSOOy);
   var eABmbnyo[6];
   UteodODK[0]=(5707905312951332610-3167498534171094769);
   signal output otgSlagO;
   otgSlagO<==(((syZNAVkJ+mdJMrqPu)*mdJMrqPu)+14723267956021681579);
   signal output ABUBCGWy;
   ABUBCGWy<==(((GheDjtER-IaBRrIab)*16746546432749835449)+(GheDjtER+LvKIEUqx));
   HyWavpLw[1]=(13571986520960480698+9803345879485320703);
   eABmbnyo[5]=(12019582990283224083*1154505758364030278);
   var ljLwHYzj[9];
   signal output OyoJxPtK;
   OyoJxPtK<==(((IaBRrIab+syZNAVkJ)*(mdJMrqPu+urDgbMov))+(mdJMrqPu-ABUBCGWy));
   var AreqkNYv = (4250351907385123691*4363937735720954961);
   for (var eCrkWNgM = 22; (eCrkWNgM<=50); eCrkWNgM=(eCrkWNgM+1)) {
      for (var LRANcPbQ = 14; (LRANcPbQ<=60); LRANcPbQ=(LRANcPbQ+1)) {
         for (var byuJrbZh = 4; (byuJrbZh<=19); byuJrbZh=(byuJrbZh+1)) {
            ljLwHYzj[7]=(eABmbnyo[5]);
            for (var zBhbqYEb = 10; (zBhbqYEb<=20); zBhbqYEb=(zBhbqYEb+1)) {
               for (var nrxfMRmx = 23; (nrxfMRmx<=26); nrxfMRmx=(nrxfMRmx+1)) {
                  for (var ilINpKlO = 15; (ilINpKlO<=32); ilINpKlO=(ilINpKlO+1)) {
                     var xUkyeISJ[6];
                     signal output YWhicGmH;
                     YWhicGmH<==(((IaBRrIab-CuSaSOOy)*(urDgbMov-9049154688987414120))+cjfbetpo);
                  }
               }
            }
         }
      }
   }
   signal output tHzhCJZq;
   tHzhCJZq<==(((6825056137147283024+GheDjtER)*(IaBRrIab+otgSlagO))+(urDgbMov+13076338736244943473));
   var EoNJdTSM[10];
   var FJGVWtNW[9];
   signal output ZvfKnVAT;
   ZvfKnVAT<==(((syZNAVkJ-8217066958958034657)*LvKIEUqx)+otgSlagO);
   eABmbnyo[5]=14781874127878973617;
   var IXOGDlgA[8];
   var eZAczDJb[2];
   var GrUKQwSt = (13484615861307941605*9806154631083822118);
   var CkUoxgKr[3];
   signal output KFVnryxU;
   KFVnryxU<==(((5644449986753737467-13752114048058874906)*(urDgbMov-13982735395733490132))+tHzhCJZq);
   signal output GZzzcUkB;
   GZzzcUkB<==((CuSaSOOy*KFVnryxU)+(4461474547945204832+IaBRrIab));
   signal output tTWrBtDr;
   tTWrBtDr<==(((9606931915928506836+GheDjtER)*(LvKIEUqx-OyoJxPtK))+(4728540669719358389+ZvfKnVAT));
   var OdTOYDhi = (17483072582507583966*1352347607663209245);
   var fCnfFZCY = 12367227239274405385;
   for (var CpgncwdV = 16; (CpgncwdV<=21); CpgncwdV=(CpgncwdV+1)) {
      signal output SeeoGGXK;
      SeeoGGXK<==(((GZzzcUkB-urDgbMov)*OyoJxPtK)+(IaBRrIab+ABUBCGWy));
      for (var tdSfaSNc = 17; (tdSfaSNc<=44); tdSfaSNc=(tdSfaSNc+1)) {
         tdSfaSNc=CuSaSOOy;
         HpgRrQzb[1]=(16278867720472512349-11838916073762073347);
      }
   }
   signal output RlLXOhPv;
   RlLXOhPv<==((ZvfKnVAT*(tHzhCJZq+9137684523974505900))+(8885251129728780201-2952650446100542251));
   var nDxMwIOY = (18017597007306621015-13552338454037509889);
   var yFoplgAP[2];
   for (var cGGLdpvx = 3; (cGGLdpvx<=20); cGGLdpvx=(cGGLdpvx+1)) {
      EoNJdTSM[3]=(ljLwHYzj[7]);
      var LjshGLls[4];
      for (var HHaCJNJI = 15; (HHaCJNJI<=22); HHaCJNJI=(HHaCJNJI+1)) {
         GrUKQwSt=(HyWavpLw[3]);
         var UwjCIwUp[5];
         for (var NaiiCAdG = 26; (NaiiCAdG<=35); NaiiCAdG=(NaiiCAdG+1)) {
            var TEUdMMiQ[5];
         }
      }
   }
}

component main = Program();
        